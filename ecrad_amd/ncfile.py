"""Minimal netCDF access for the host side: classic files (CDF-1/CDF-2) through scipy, netCDF-4 / HDF5 files through the C layer of the Fortran host.

The reference reads every table and input through utilities/easy_netcdf.F90, which hands
``real(jprb)`` arrays to the netCDF library and lets it widen float32 -> double.  We do the same:
every numeric variable comes back as float64 (or int32), in the *file's* C index order.  A C-ordered
numpy array of shape (d0, d1, d2) is byte-identical to the Fortran array (d2, d1, d0) that
easy_netcdf returns without transposition (utilities/easy_netcdf.F90:1040-1150), so "first Fortran
index fastest" == "last numpy index fastest" everywhere in this package.
"""
from __future__ import annotations

import numpy as np
from scipy.io import netcdf_file


class _Var4:
    """What NcFile uses of a scipy netcdf variable, for a variable of a netCDF-4 file."""

    def __init__(self, owner, varid, name, xtype, dims):
        self._owner, self._varid, self.name, self._xtype = owner, varid, name, xtype
        self.dimensions = tuple(dims)
        self.shape = tuple(owner.dimensions[d] for d in dims)

    @property
    def data(self):
        return self._owner.read(self._varid, self._xtype, self.shape)

    def __getitem__(self, key):
        return self.data[key]

    def getValue(self):
        return self.data.reshape(-1)[0]


class _Nc4:
    """A netCDF-4 / HDF5 file through the C layer of the Fortran host (ecrad_amd/fortran/nc_classic.c built as libecnc.so, which reads such files
    through the HDF5 library it loads at run time): the same reader for both hosts.  Presents the few members of scipy's netcdf_file that NcFile
    uses (`variables`, `dimensions`, global attributes as Python attributes)."""
    _T = {1: np.int8, 2: "S1", 3: np.int16, 4: np.int32, 5: np.float32, 6: np.float64}

    def __init__(self, path: str):
        import ctypes as C
        import os
        lib_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fortran", "libecnc.so")
        if not os.path.exists(lib_path):
            raise OSError(f"{path} is a netCDF-4 / HDF5 file; reading it needs {lib_path} (make -C ecrad_amd/fortran libecnc.so) and an HDF5 library")
        self._C, self._lib = C, C.CDLL(lib_path)
        lib = self._lib
        lib.ecnc_strerror.restype = C.c_char_p
        ncid = C.c_int()
        st = lib.ecnc_open(path.encode(), C.byref(ncid))
        if st != 0:
            raise OSError(f"{path}: {lib.ecnc_strerror(st).decode()}")
        self._ncid = ncid.value
        nd, nv, ng, ul = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        lib.ecnc_inq(self._ncid, C.byref(nd), C.byref(nv), C.byref(ng), C.byref(ul))
        name = C.create_string_buffer(256)
        dim_names, self.dimensions = [], {}
        for d in range(nd.value):
            n = C.c_longlong()
            lib.ecnc_inq_dim(self._ncid, d, name, 256, C.byref(n))
            dim_names.append(name.value.decode())
            self.dimensions[dim_names[-1]] = int(n.value)
        self.variables = {}
        for v in range(nv.value):
            xt, rk, na = C.c_int(), C.c_int(), C.c_int()
            dimids = (C.c_int * 8)()
            lib.ecnc_inq_var(self._ncid, v, name, 256, C.byref(xt), C.byref(rk), dimids, C.byref(na))
            vn = name.value.decode()
            var = _Var4(self, v, vn, xt.value, [dim_names[dimids[k]] for k in range(rk.value)])
            for k in range(na.value):
                lib.ecnc_inq_attname(self._ncid, v, k, name, 256)
                setattr(var, name.value.decode(), self._att(v, name.value))
            self.variables[vn] = var
        for k in range(ng.value):
            lib.ecnc_inq_attname(self._ncid, -1, k, name, 256)
            if not hasattr(self, name.value.decode()):
                setattr(self, name.value.decode(), self._att(-1, name.value))

    def _att(self, varid, name):
        C, lib = self._C, self._lib
        xt, n = C.c_int(), C.c_longlong()
        lib.ecnc_inq_att(self._ncid, varid, name, C.byref(xt), C.byref(n))
        if xt.value == 2:
            buf = C.create_string_buffer(int(n.value) + 1)
            lib.ecnc_get_att(self._ncid, varid, name, 2, buf, n, None)
            return buf.raw[:int(n.value)]
        out = np.empty(int(n.value), dtype=self._T[xt.value])
        lib.ecnc_get_att(self._ncid, varid, name, xt.value, out.ctypes.data_as(C.c_void_p), n, None)
        return out[0] if out.size == 1 else out

    def read(self, varid, xtype, shape):
        C = self._C
        out = np.empty(shape, dtype=self._T[xtype])
        st = self._lib.ecnc_get_vara(self._ncid, varid, xtype, out.ctypes.data_as(C.c_void_p), 0, None, None)
        if st != 0:
            raise OSError(self._lib.ecnc_strerror(st).decode())
        return out

    def close(self):
        if self._ncid is not None:
            self._lib.ecnc_close(self._ncid)
            self._ncid = None


class NcFile:
    def __init__(self, path: str):
        self.path = path
        with open(path, "rb") as fh:
            magic = fh.read(4)
        # (netCDF-4 files are HDF5 files: signature \211HDF; every file of the reference is classic, a user's input need not be)
        self._f = _Nc4(path) if magic == b"\x89HDF" else netcdf_file(path, "r", mmap=False)

    def exists(self, name: str) -> bool:
        return name in self._f.variables

    def rank(self, name: str) -> int:
        """file%get_rank: -1 if absent (utilities/easy_netcdf.F90 get_rank)."""
        if name not in self._f.variables:
            return -1
        return len(self._f.variables[name].shape)

    def get(self, name: str) -> np.ndarray:
        v = self._f.variables[name]
        a = np.array(v.data)
        if a.dtype.kind == "f":
            a = a.astype(np.float64)
        elif a.dtype.kind in "iu":
            a = a.astype(np.int64)
        return a

    def get_scalar(self, name: str) -> float:
        return float(np.asarray(self.get(name)).reshape(-1)[0])

    def global_attr(self, name: str) -> str:
        v = getattr(self._f, name)
        if isinstance(v, bytes):
            v = v.decode("utf-8", "replace")
        return str(v)

    def dims(self):
        return dict(self._f.dimensions)

    def close(self):
        self._f.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def write_nc(path: str, dims: dict, variables: dict, attrs: dict | None = None,
             double: bool = True) -> None:
    """Write a classic netCDF file.  ``variables[name] = (dim_names, array[, var_attrs])``.

    Counterpart of radiation_save.F90's use of easy_netcdf for the offline driver's outputs.
    """
    # (64-bit offsets -- CDF-2, what the reference's easy_netcdf creates for large files -- when the data do not fit CDF-1's 2 GiB)
    nbytes = sum(np.asarray(spec[1]).size * (8 if double or np.asarray(spec[1]).dtype.kind != "f" else 4) for spec in variables.values())
    f = netcdf_file(path, "w", version=2 if nbytes > 2**31 - 2**24 else 1)
    for k, v in (attrs or {}).items():
        setattr(f, k, v)
    for d, n in dims.items():
        f.createDimension(d, n)
    for name, spec in variables.items():
        dim_names, arr = spec[0], np.asarray(spec[1])
        vattrs = spec[2] if len(spec) > 2 else {}
        if arr.dtype.kind == "f":
            typ = "d" if double else "f"
        else:
            typ = "i"
        v = f.createVariable(name, typ, tuple(dim_names))
        for ak, av in vattrs.items():
            setattr(v, ak, av)
        if arr.ndim == 0:
            v.data[...] = arr
        else:
            v[:] = arr
    f.close()
