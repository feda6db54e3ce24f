#!/usr/bin/env python3
"""oracle/gen_rrtm_dumper.py -- TEST / DATA-PREPARATION INFRASTRUCTURE (own code).

Writes oracle/_ref/rrtm/dump_rrtm_tables.F90: a program that `use`s the reference's RRTMG table modules
(ifsrrtm/yoerrta1-16, yoesrta16-29, yoerrtrf, yoerrtwn, yoesrtwn, yoerrtftr, yoesrtm), as compiled from
where they lie by oracle/build_ref_rrtm.sh, runs the reference's own initialisation (ref_rrtm_setup:
SURRTAB/SURRTPK/SURRTRF/RRTM_INIT_140GP/SRTM_INIT, which read data/RADRRTM and data/RADSRTM and reduce
them to 140/112 g-points) and writes every table the gas-optics routines read to one binary file.
oracle/make_rrtm_tables.py turns that file into data/rrtmg_tables.npz ("dumps of RRTMG
module tables").  The list of arrays is taken from the declarations in the module sources, so nothing of
the reference is copied: only names and shapes are read here, values come out of the running library.
"""
import re, sys, os

REF = os.environ.get("REF", "/root/reference")
out = sys.argv[1]
mods = [f"yoerrta{i}" for i in range(1, 17)] + [f"yoesrta{i}" for i in range(16, 30)] + \
       ["yoerrtrf", "yoerrtwn", "yoesrtwn", "yoerrtftr", "yoesrtm"]
# unreduced (16 g-points per band) shortwave tables and double-precision reading buffers are not used by
# the gas-optics routines; KA/KB/KAC/KBC are EQUIVALENCEd views of ABSA/ABSB
skip = re.compile(r"^(KA|KB|KAC|KBC|K[AB]_D|.*_D|SELFREF|FORREF|SFLUXREF|RAYLA|RAYLB|ABSO3A|ABSO3B|ABSCH4|ABSH2O|ABSCO2)$")
lines = ["program dump_rrtm_tables", "  use ref_rrtm_wrappers, only : ref_rrtm_setup", "  use iso_c_binding"]
calls = []
for m in mods:
    src = open(f"{REF}/ifsrrtm/{m}.F90").read()
    src = re.sub(r"&\s*\n\s*&?", "", src)
    names = []
    for ln in src.splitlines():
        ln = ln.split("!")[0].strip()
        mm = re.match(r"(REAL|INTEGER)\s*\(KIND=(JPRB|JPIM)\)\s*(,\s*PARAMETER)?\s*(,\s*DIMENSION\s*\([^)]*\))?\s*::\s*(.*)$", ln, re.I)
        if not mm or mm.group(3):
            continue
        has_dim = mm.group(4) is not None
        kind = "r" if mm.group(1).upper() == "REAL" else "i"
        # split the declaration list at top-level commas
        depth, cur, items = 0, "", []
        for ch in mm.group(5):
            if ch == "(": depth += 1
            if ch == ")": depth -= 1
            if ch == "," and depth == 0: items.append(cur); cur = ""
            else: cur += ch
        items.append(cur)
        for it in items:
            it = it.strip()
            nm = re.match(r"([A-Za-z_0-9]+)", it).group(1)
            is_sw = m.startswith("yoesrta")
            if skip.match(nm.upper()) and (is_sw or nm.upper() in ("KA", "KB")):
                continue
            is_array = has_dim or "(" in it
            names.append((nm, kind, is_array))
    if not names:
        continue
    only = ", ".join(f"{m}_{n} => {n}" for n, _, _ in names)
    lines.append(f"  use {m}, only : {only}")
    for n, kind, is_array in names:
        v = f"{m}_{n}"
        if is_array:
            calls.append(f"  call dump{kind}('{m}.{n.lower()}', {len(m)+1+len(n)}, size(shape({v})), shape({v}), {v})")
        else:
            calls.append(f"  call dump{kind}('{m}.{n.lower()}', {len(m)+1+len(n)}, 0, [1], [{v}])")
lines += ["  implicit none", "  external :: dumpr, dumpi", "  character(len=512) :: dir, outfile", "  integer :: n",
          "  call get_command_argument(1, dir)", "  call get_command_argument(2, outfile)",
          "  n = len_trim(dir)", "  call ref_rrtm_setup(trim(dir)//c_null_char, n)",
          "  open(unit=77, file=trim(outfile), access='stream', form='unformatted', status='replace')"]
lines += calls
lines += ["  close(77)", "end program", "",
          "subroutine dumpr(name, nchar, nd, shp, a)", "  use parkind1, only : jprb", "  implicit none",
          "  integer :: nchar, nd, shp(*), i, n", "  character(len=*) :: name", "  real(jprb) :: a(*)",
          "  n = 1", "  do i = 1, nd", "    n = n * shp(i)", "  end do",
          "  write(77) nchar, name(1:nchar), 8, nd, (shp(i), i = 1, nd)", "  write(77) (real(a(i), 8), i = 1, n)",
          "end subroutine", "",
          "subroutine dumpi(name, nchar, nd, shp, a)", "  implicit none",
          "  integer :: nchar, nd, shp(*), i, n, a(*)", "  character(len=*) :: name",
          "  n = 1", "  do i = 1, nd", "    n = n * shp(i)", "  end do",
          "  write(77) nchar, name(1:nchar), 4, nd, (shp(i), i = 1, nd)", "  write(77) (a(i), i = 1, n)",
          "end subroutine"]
open(out, "w").write("\n".join(lines) + "\n")
print(f"{len(calls)} arrays -> {out}")
