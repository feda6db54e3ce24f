"""netCDF-4 / HDF5 INPUT files (utilities/easy_netcdf.F90:133-200 opens whatever the netCDF library opens, and a user's
input file is as likely netCDF-4 as classic).  Both hosts read them through ecrad_amd/fortran/nc_classic.c, which loads the HDF5 library at run time
and applies netCDF-4's conventions (dimension scales, DIMENSION_LIST, _Netcdf4Dimid, hidden attributes):
  * a file written by the HDF5 library's OWN high-level API (H5DS dimension scales, H5LT attributes: what libnetcdf itself calls) -- independent of
    every writer of this repository -- read by the Python host (ncfile.NcFile) and by the Fortran `netcdf` module;
  * the files of the repository's two writers read back;
  * the reference's input file test/ifs/ecrad_meridian.nc converted to netCDF-4 gives the driver the same input arrays."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from ecrad_amd.ncfile import NcFile
from test_fortran_netcdf import exe      # noqa: F401  (fixture: builds tests/_build/nctest/nc_roundtrip from netcdf.F90 + nc_classic.c)
from test_hdf5_output import H5

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MERIDIAN = os.path.join(ROOT, "tests", "golden", "ecrad_meridian.nc")
LIBECNC = os.path.join(ROOT, "ecrad_amd", "fortran", "libecnc.so")
pytestmark = pytest.mark.skipif(not os.path.exists(LIBECNC), reason="ecrad_amd/fortran/libecnc.so has not been built (make -C ecrad_amd/fortran)")


def _write_with_the_hdf5_library(path):
    """level(3) with a coordinate variable, column(4) as a pure dimension, a(level, column) double with attributes, i(column) int32, s scalar,
    u(time unlimited = 2, column) float in chunks, a global text attribute: through H5DSset_scale / H5DSattach_scale / H5LTset_attribute_*."""
    h = H5()
    h5, hl, hid = h.h5, h.hl, h.hid
    h5.H5Fcreate.restype = hid; h5.H5Fcreate.argtypes = [C.c_char_p, C.c_uint, hid, hid]
    h5.H5Screate_simple.restype = hid; h5.H5Screate_simple.argtypes = [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    h5.H5Screate.restype = hid; h5.H5Screate.argtypes = [C.c_int]
    h5.H5Dcreate2.restype = hid; h5.H5Dcreate2.argtypes = [hid, C.c_char_p, hid, hid, hid, hid, hid]
    h5.H5Dwrite.argtypes = [hid, hid, hid, hid, hid, C.c_void_p]
    h5.H5Pcreate.restype = hid; h5.H5Pcreate.argtypes = [hid]
    h5.H5Pset_chunk.argtypes = [hid, C.c_int, C.POINTER(C.c_uint64)]
    hl.H5DSset_scale.argtypes = [hid, C.c_char_p]
    hl.H5DSattach_scale.argtypes = [hid, hid, C.c_uint]
    hl.H5LTset_attribute_string.argtypes = [hid, C.c_char_p, C.c_char_p, C.c_char_p]
    hl.H5LTset_attribute_double.argtypes = [hid, C.c_char_p, C.c_char_p, C.POINTER(C.c_double), C.c_size_t]
    hl.H5LTset_attribute_int.argtypes = [hid, C.c_char_p, C.c_char_p, C.POINTER(C.c_int), C.c_size_t]
    NATIVE_FLOAT = hid.in_dll(h5, "H5T_NATIVE_FLOAT_g").value
    IEEE_F64LE = hid.in_dll(h5, "H5T_IEEE_F64LE_g").value
    IEEE_F32LE = hid.in_dll(h5, "H5T_IEEE_F32LE_g").value
    STD_I32LE = hid.in_dll(h5, "H5T_STD_I32LE_g").value
    DCPL = hid.in_dll(h5, "H5P_CLS_DATASET_CREATE_ID_g").value
    f = h5.H5Fcreate(path.encode(), 2, 0, 0)      # H5F_ACC_TRUNC
    assert f >= 0

    def space(*shape, maxshape=None):
        if not shape:
            return h5.H5Screate(0)
        dims = (C.c_uint64 * len(shape))(*shape)
        mx = (C.c_uint64 * len(shape))(*maxshape) if maxshape else None
        return h5.H5Screate_simple(len(shape), dims, mx)

    def dataset(name, ftype, mtype, arr, shape, dcpl=0, maxshape=None):
        d = h5.H5Dcreate2(f, name.encode(), ftype, space(*shape, maxshape=maxshape), 0, dcpl, 0)
        assert d >= 0, name
        if arr is not None:
            a = np.ascontiguousarray(arr)
            assert h5.H5Dwrite(d, mtype, 0, 0, 0, a.ctypes.data) >= 0
        return d
    level = dataset("level", IEEE_F64LE, h.NATIVE_DOUBLE, np.array([1000.0, 500.0, 10.0]), (3,))
    column = dataset("column", IEEE_F32LE, NATIVE_FLOAT, None, (4,))
    a = 10.0 * np.arange(1, 4)[:, None] + np.arange(1, 5)[None, :] + 0.125
    da = dataset("a", IEEE_F64LE, h.NATIVE_DOUBLE, a, (3, 4))
    di = dataset("i", STD_I32LE, h.NATIVE_INT, np.array([7, -8, 9, 2147483647], dtype=np.int32), (4,))
    ds = dataset("s", IEEE_F64LE, h.NATIVE_DOUBLE, np.array(3.141592653589793), ())
    dcpl = h5.H5Pcreate(DCPL)
    h5.H5Pset_chunk(dcpl, 2, (C.c_uint64 * 2)(1, 4))
    unlimited = 0xFFFFFFFFFFFFFFFF
    dcpl1 = h5.H5Pcreate(DCPL)
    h5.H5Pset_chunk(dcpl1, 1, (C.c_uint64 * 1)(1))
    time = dataset("time", IEEE_F32LE, NATIVE_FLOAT, None, (2,), dcpl=dcpl1, maxshape=(unlimited,))
    u = np.array([[1.5, -2.25, 3.0, 1.0e10], [0.5, 0.25, 0.125, 0.0625]], dtype=np.float32)
    du = dataset("u", IEEE_F32LE, NATIVE_FLOAT, u, (2, 4), dcpl=dcpl, maxshape=(unlimited, 4))
    # netCDF-4's conventions, by the library calls libnetcdf makes
    for k, (d, name, pure, n) in enumerate(((level, "level", False, 3), (column, "column", True, 4), (time, "time", True, 2))):
        text = ("This is a netCDF dimension but not a netCDF variable.%10d" % n) if pure else name
        assert hl.H5DSset_scale(d, text.encode()) >= 0
        assert hl.H5LTset_attribute_int(f, name.encode(), b"_Netcdf4Dimid", (C.c_int * 1)(k), 1) >= 0
    assert hl.H5DSattach_scale(da, level, 0) >= 0 and hl.H5DSattach_scale(da, column, 1) >= 0
    assert hl.H5DSattach_scale(di, column, 0) >= 0
    assert hl.H5DSattach_scale(du, time, 0) >= 0 and hl.H5DSattach_scale(du, column, 1) >= 0
    assert hl.H5LTset_attribute_string(f, b"a", b"units", b"W m-2") >= 0
    assert hl.H5LTset_attribute_double(f, b"a", b"_FillValue", (C.c_double * 1)(-999.0), 1) >= 0
    assert hl.H5LTset_attribute_double(f, b"a", b"valid_range", (C.c_double * 2)(0.0, 100.0), 2) >= 0
    assert hl.H5LTset_attribute_int(f, b"i", b"answer", (C.c_int * 1)(42), 1) >= 0
    assert hl.H5LTset_attribute_string(f, b"/", b"title", b"written by the HDF5 library") >= 0
    assert hl.H5LTset_attribute_string(f, b"/", b"_NCProperties", b"version=2,hdf5=1.10") >= 0
    for d in (level, column, da, di, ds, time, du):
        h5.H5Dclose(d)
    h5.H5Fclose(f)
    return a, u


def test_python_host_reads_a_file_written_by_the_hdf5_library(tmp_path):
    path = str(tmp_path / "hl.nc")
    a, u = _write_with_the_hdf5_library(path)
    assert open(path, "rb").read(4) == b"\x89HDF"
    with NcFile(path) as nc:
        assert nc.dims() == {"level": 3, "column": 4, "time": 2}
        assert nc.exists("a") and nc.exists("level") and not nc.exists("column") and not nc.exists("time")      # pure dimensions are not variables
        assert nc.rank("a") == 2 and nc.rank("s") == 0 and nc.rank("nothing") == -1
        assert nc._f.variables["a"].dimensions == ("level", "column") and nc._f.variables["u"].dimensions == ("time", "column")
        assert np.array_equal(nc.get("a"), a) and nc.get("a").dtype == np.float64
        assert np.array_equal(nc.get("u"), u.astype(np.float64))
        assert list(nc.get("i")) == [7, -8, 9, 2147483647]
        assert nc.get_scalar("s") == 3.141592653589793
        assert list(nc.get("level")) == [1000.0, 500.0, 10.0]
        va = nc._f.variables["a"]
        assert va.units == b"W m-2" and va._FillValue == -999.0 and list(va.valid_range) == [0.0, 100.0]
        assert nc._f.variables["i"].answer == 42
        assert nc.global_attr("title") == "written by the HDF5 library"
        for hidden in ("DIMENSION_LIST", "CLASS", "NAME", "REFERENCE_LIST", "_Netcdf4Dimid"):
            assert not hasattr(va, hidden) and not hasattr(nc._f.variables["level"], hidden)
        assert not hasattr(nc._f, "_NCProperties")


def test_fortran_module_reads_the_same_file(exe, tmp_path):      # noqa: F811
    build = exe
    path = str(tmp_path / "hl.nc")
    a, u = _write_with_the_hdf5_library(path)
    for var, want in (("a", a), ("u", u.astype(np.float64))):
        p = subprocess.run([build, "dump", path, var], capture_output=True, text=True)
        assert p.returncode == 0 and p.stdout.startswith("DUMP"), p.stdout + p.stderr
        t = p.stdout.split()
        assert int(t[1]) == 2 and (int(t[2]), int(t[3])) == want.shape[::-1]      # Fortran (column, level) = C (level, column)
        flat = want.reshape(-1)
        assert float(t[5]) == flat[0] and float(t[6]) == flat[-1] and abs(float(t[4]) - flat.sum()) <= 1e-12 * np.abs(flat).sum()


def test_files_of_the_two_writers_of_this_repository_read_back(exe, tmp_path):      # noqa: F811
    from ecrad_amd.hdf5file import write_nc4
    H5()      # (skips where the image has no HDF5 library: the reader loads it at run time)
    rng = np.random.default_rng(5)
    dims = {"column": 5, "half_level": 4, "band": 3}
    variables = {"flux_up": (("half_level", "column"), rng.standard_normal((4, 5)), {"units": "W m-2", "long_name": "Upwelling flux"}),
                 "band": (("band",), np.array([1.0, 2.0, 3.0])),
                 "count": (("column",), np.arange(5, dtype=np.int32)),
                 "spectral": (("half_level", "column", "band"), rng.standard_normal((4, 5, 3)))}
    path = str(tmp_path / "py.nc")
    write_nc4(path, dims, variables, attrs={"title": "python writer"})
    with NcFile(path) as nc:
        assert nc.dims() == dims
        for name, spec in variables.items():
            assert np.array_equal(nc.get(name), np.asarray(spec[1], dtype=np.float64 if np.asarray(spec[1]).dtype.kind == "f" else np.int64)), name
            assert nc._f.variables[name].dimensions == tuple(spec[0])
        assert nc._f.variables["flux_up"].units == b"W m-2" and nc.global_attr("title") == "python writer"
    build = exe
    if True:      # ... and the Fortran module's own netCDF-4 output through the Fortran module's reader
        f4 = str(tmp_path / "t4.nc")
        p = subprocess.run([build, "write_hdf5", f4], capture_output=True, text=True)
        assert p.returncode == 0 and "WRITE OK" in p.stdout, p.stdout + p.stderr
        p = subprocess.run([build, "read", f4], capture_output=True, text=True)
        assert p.returncode == 0 and "READ OK" in p.stdout, p.stdout + p.stderr
        with NcFile(f4) as nc:
            assert nc.dims() == {"column": 4, "level": 3, "five": 5}
            assert np.array_equal(nc.get("a"), 10.0 * np.arange(1, 4)[:, None] + np.arange(1, 5)[None, :] + 0.125)


def test_the_reference_input_file_as_netcdf4_gives_the_driver_the_same_arrays(tmp_path):
    """test/ifs/ecrad_meridian.nc (classic, `column` as record dimension) rewritten as netCDF-4 by the repository's writer: read_input
    (driver/ecrad_driver_read_input.F90 restated in ecrad_amd/driver.py) builds identical inputs from the two files."""
    from scipy.io import netcdf_file
    from ecrad_amd.cases import make_config
    from ecrad_amd.driver import DriverConfig, read_input
    from ecrad_amd.cases import NAMELIST
    from ecrad_amd.hdf5file import write_nc4
    H5()      # (skips where the image has no HDF5 library)
    with netcdf_file(MERIDIAN, "r", mmap=False) as nc:
        dims = {d: (n if n is not None else nc.variables["pressure_hl"].shape[0]) for d, n in nc.dimensions.items()}
        variables = {name: (v.dimensions, np.array(v.data).reshape(v.shape), {k: getattr(v, k).decode() for k in ("units",) if hasattr(v, k)}) for name, v in nc.variables.items()}
    path = str(tmp_path / "meridian4.nc")
    write_nc4(path, dims, variables, double=True)      # (float32 values widen exactly; `iseed` is a double in the file)
    config = make_config("Tripleclouds")
    dc = DriverConfig.read(NAMELIST)
    a = read_input(MERIDIAN, config, dc)
    b = read_input(path, config, dc)
    assert a[0] == b[0] and a[1] == b[1]
    for oa, ob in zip(a[2:], b[2:]):
        if oa is None:
            assert ob is None
            continue
        for k, va in vars(oa).items():
            vb = getattr(ob, k)
            if isinstance(va, np.ndarray):
                assert np.array_equal(va, vb, equal_nan=True), k
