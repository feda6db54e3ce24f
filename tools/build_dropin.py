#!/usr/bin/env python
"""Build the reference's OWN offline driver with the GPU drop-in in place of its radiation_interface module.

    python tools/build_dropin.py [--ref /root/reference] [--out tests/_build/dropin] [-j 8]

What is compiled, all with amdflang, objects and .mod files only under --out (git-ignored; the binary travels to the GPU box):
  * from where they lie under --ref, unmodified: ifsaux/, drhook/yomhook_dummy.F90, utilities/, ifsrrtm/, radiation/ EXCEPT
    radiation_interface.F90, and of driver/ the program ecrad_driver.F90 with ecrad_driver_config / _read_input /
    print_matrix_mod -- the host side the north star wants kept: namelist, config_type, table preparation, netCDF I/O;
  * from this repo: ecrad_amd/fortran/netcdf.F90 + nc_classic.c (the `netcdf` module the reference's easy_netcdf.F90 is
    written against, over classic-format files: this image has no libnetcdff), ecrad_hip_binding.F90,
    radiation_hip_interface.F90 and radiation_hip_rrtmg.F90 with -DECRAD_HIP_REFERENCE_TYPES, and
    ecrad_amd/fortran/radiation_interface.F90 -- the drop-in module of the reference's name;
  * linked against ecrad_amd/csrc/libecrad_hip.so (rpath relative to the binary) -> <out>/ecrad_hip.
Without --openmp the driver's loop over blocks runs serially (what a GPU host wants: one handle per process, large blocks,
INTEGRATION.md 1.3); with --openmp the PARALLEL DO of driver/ecrad_driver.F90:348 is live and the threads' calls of radiation() queue
in the drop-in (-> <out>_omp, the re-entrancy test).

This is BOUNDARY PROOF (the drop-in executed inside its host), not an oracle pin: the reference's Fortran here only prepares
tables and moves files; every flux comes from the HIP library.  Nothing of the reference is copied into the repo."""
import argparse
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FC = "/opt/rocm/bin/amdflang"
CC = "/opt/rocm/lib/llvm/bin/clang"


def sources(ref, reference_build=False):
    src = []
    for d in ("ifsaux", "utilities", "ifsrrtm", "radiation"):
        for f in sorted(os.listdir(os.path.join(ref, d))):
            # (easy_netcdf_read_mpi.F90 belongs to builds with the IFS's FIAT library only: utilities/CMakeLists.txt:17)
            if f.endswith(".F90") and not (d == "radiation" and f == "radiation_interface.F90" and not reference_build) and f != "easy_netcdf_read_mpi.F90":
                src.append(os.path.join(ref, d, f))
    src.append(os.path.join(ref, "drhook", "yomhook_dummy.F90"))
    for f in ("ecrad_driver_config.F90", "ecrad_driver_read_input.F90", "print_matrix_mod.F90"):
        src.append(os.path.join(ref, "driver", f))
    # the IFS-side caller (ifs/radiation_scheme.F90:540 is the second call site of radiation()) and its blocking
    for f in sorted(os.listdir(os.path.join(ref, "ifs"))):
        if f.endswith(".F90") and f != "cos_sza.F90":       # (not among the SOURCES of ifs/Makefile)
            src.append(os.path.join(ref, "ifs", f))
    src.append(os.path.join(ref, "driver", "ifs_blocking.F90"))
    ours = os.path.join(ROOT, "ecrad_amd", "fortran")
    mine = ("netcdf.F90",) if reference_build else ("netcdf.F90", "ecrad_hip_binding.F90", "radiation_hip_interface.F90", "radiation_hip_rrtmg.F90", "radiation_interface.F90")
    for f in mine:
        src.append(os.path.join(ours, f))
    return src


# executable -> the file of driver/ that holds its main program
PROGRAMS = {"ecrad": "ecrad_driver.F90", "ecrad_ifs": "ecrad_ifs_driver.F90", "ecrad_ifs_blocked": "ecrad_ifs_driver_blocked.F90"}


def scan(path):
    """(modules defined, modules used) by a source file."""
    defines, uses = set(), set()
    for line in open(path, errors="replace"):
        s = line.split("!")[0].strip().lower()
        m = re.match(r"module\s+(\w+)\s*$", s)
        if m and m.group(1) != "procedure":
            defines.add(m.group(1))
        m = re.match(r"use\s*(?:,\s*(?:non_)?intrinsic\s*)?(?:::)?\s*(\w+)", s)
        if m:
            uses.add(m.group(1))
    return defines, uses


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "_build", "dropin"))
    ap.add_argument("-j", type=int, default=8)
    ap.add_argument("--reference", action="store_true",
                    help="the UNMODIFIED reference instead (its own radiation_interface.F90, OpenMP on, no GPU library): the CPU "
                         "executable `ecrad_ref`, whose only non-reference part is the netCDF library underneath easy_netcdf.F90")
    ap.add_argument("--single", action="store_true", help="the host in single precision (-DPARKIND1_SINGLE: jprb = real32, as the IFS runs)")
    ap.add_argument("--openmp", action="store_true",
                    help="the drop-in compiled with OpenMP: the driver's PARALLEL DO over blocks (driver/ecrad_driver.F90:348) is live and "
                         "several host threads call radiation() at once (-> <out>_omp)")
    args = ap.parse_args()
    if args.reference and args.out.endswith("dropin"):
        args.out = os.path.join(ROOT, "tests", "_build", "reference")
    if args.single:
        args.out += "_sp"
    if args.openmp:
        args.out += "_omp"
    out, obj = os.path.abspath(args.out), os.path.join(os.path.abspath(args.out), "obj")
    os.makedirs(obj, exist_ok=True)
    lib = sources(args.ref, args.reference)
    mains = {exe: os.path.join(args.ref, "driver", f) for exe, f in PROGRAMS.items()}
    src = lib + list(mains.values())
    info = {f: scan(f) for f in src}
    owner = {}
    for f, (d, _) in info.items():
        for m in d:
            owner[m] = f
    deps = {f: {owner[m] for m in u if m in owner and owner[m] != f} for f, (_, u) in info.items()}
    flags = (["-O3", "-fopenmp", "-fPIC", "-cpp"] if args.reference else ["-O1", "-fPIC", "-cpp", "-DECRAD_HIP_REFERENCE_TYPES"]) + [ f"-I{args.ref}/include", f"-I{args.ref}/radiation",
             f"-I{args.ref}/ifsaux", f"-I{args.ref}/ifsrrtm", f"-I{args.ref}/ifs", f"-I{obj}", "-module-dir", obj]
    # (utilities/easy_netcdf.F90:232-240: the reference asks for NF90_HDF5 -- is_hdf5_file, the driver's do_write_hdf5 -- only when it is
    #  compiled with NC_NETCDF4, as its own Makefile does where the netCDF library has netCDF-4; this repo's netcdf module writes that format)
    flags.append("-DNC_NETCDF4")
    if args.single:
        flags.append("-DPARKIND1_SINGLE")
    if args.openmp and not args.reference:
        flags.append("-fopenmp")

    def obj_of(f):
        return os.path.join(obj, os.path.basename(f)[:-4] + ".o")

    def compile_one(f):
        o = obj_of(f)
        if os.path.exists(o) and os.path.getmtime(o) >= max([os.path.getmtime(f)] + [os.path.getmtime(obj_of(d)) for d in deps[f] if os.path.exists(obj_of(d))]):
            return f, 0, ""
        p = subprocess.run([FC, *flags, "-c", f, "-o", o], capture_output=True, text=True)
        return f, p.returncode, p.stderr[-3000:]

    done, todo = set(), set(src)
    with ThreadPoolExecutor(max_workers=args.j) as ex:
        while todo:
            ready = [f for f in todo if deps[f] <= done]
            if not ready:
                sys.exit("dependency cycle among: " + ", ".join(sorted(os.path.basename(f) for f in todo)))
            for f, rc, err in ex.map(compile_one, ready):
                if rc != 0:
                    sys.exit(f"{f} failed:\n{err}")
                done.add(f)
                todo.discard(f)
    nc_o = os.path.join(obj, "nc_classic.o")
    subprocess.run([CC, "-O2", "-fPIC", "-c", os.path.join(ROOT, "ecrad_amd", "fortran", "nc_classic.c"), "-o", nc_o], check=True)
    csrc = os.path.join(ROOT, "ecrad_amd", "csrc")
    rel = os.path.relpath(csrc, out)
    for name, main_src in mains.items():
        exe = os.path.join(out, (name + "_ref") if args.reference else (name + "_hip"))
        objs = [obj_of(f) for f in lib] + [obj_of(main_src), nc_o]
        if args.reference:
            cmd = [FC, "-fopenmp", "-o", exe, *objs]
        else:
            # (-fopenmp at the link only: the drivers call omp_get_wtime / omp_get_thread_num unconditionally; the sources are
            #  compiled without it, so their loops over blocks are serial)
            cmd = [FC, "-fopenmp", "-o", exe, *objs, f"-L{csrc}", "-lecrad_hip", f"-Wl,-rpath,$ORIGIN/{rel}"]
        p = subprocess.run(cmd, capture_output=True, text=True)
        if p.returncode != 0:
            sys.exit(f"link of {exe} failed:\n" + p.stderr[-4000:])
        print("built", exe)


if __name__ == "__main__":
    main()
