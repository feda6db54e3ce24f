"""Minimal classic-netCDF (CDF-1/CDF-2) access for the host side.

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


class NcFile:
    def __init__(self, path: str):
        self.path = path
        self._f = netcdf_file(path, "r", mmap=False)

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
