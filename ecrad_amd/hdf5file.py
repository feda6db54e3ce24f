"""A netCDF-4 (HDF5) writer for the offline drivers' output files, without the netCDF or HDF5 libraries.

The reference writes classic netCDF by default and netCDF-4/HDF5 when asked (``is_hdf5_file``: utilities/easy_netcdf.F90:212-245
-> nf90_create(..., NF90_HDF5); driver namelist ``do_write_hdf5``, driver/ecrad_driver_config.F90).  This image has neither
library on the GPU box, so the format is produced directly: an HDF5 file restricted to what such an output file needs --

  superblock version 0, one (root) group in the original symbol-table form (B-tree node + local heap + one symbol node),
  version-1 object headers, contiguous little-endian datasets (float32 / float64 / int32), attributes stored in the object
  headers, one global heap collection for the variable-length DIMENSION_LIST attributes

-- plus the conventions that make an HDF5 file a netCDF-4 file (what libnetcdf's nc4hdf.c writes): every dimension is a
"dimension scale" dataset (CLASS, NAME = "This is a netCDF dimension but not a netCDF variable.<size>", _Netcdf4Dimid,
REFERENCE_LIST), every variable carries DIMENSION_LIST (object references to its dimensions) and _Netcdf4Coordinates (their
dimension ids), and the root group carries _NCProperties.  Everything is laid out in one pass over precomputed sizes; files
are small (flux profiles of one driver run).  tests/test_hdf5_output.py reads the result back with the HDF5 library's own
tools where they exist (h5dump / libhdf5_hl's H5DS API in this container).
"""
from __future__ import annotations

import struct

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF
_SIG = b"\x89HDF\r\n\x1a\n"
_LEAF_K = 64            # symbol-table node holds 2K entries: up to 128 dimensions + variables in the root group
_INTERNAL_K = 16
NC_DIM_WITHOUT_VARIABLE = "This is a netCDF dimension but not a netCDF variable."


def _pad8(b: bytes) -> bytes:
    return b + b"\0" * (-len(b) % 8)


# ---- datatype messages ----------------------------------------------------------------------------------------------
def _dt_float(nbytes: int) -> bytes:
    if nbytes == 8:
        return struct.pack("<B3BI", 0x11, 0x20, 0x3F, 0x00, 8) + struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
    return struct.pack("<B3BI", 0x11, 0x20, 0x1F, 0x00, 4) + struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)


def _dt_int32() -> bytes:
    return struct.pack("<B3BI", 0x10, 0x08, 0x00, 0x00, 4) + struct.pack("<HH", 0, 32)


def _dt_string(n: int) -> bytes:
    return struct.pack("<B3BI", 0x13, 0x00, 0x00, 0x00, n)          # null-terminated, ASCII, fixed length n


def _dt_objref() -> bytes:
    return struct.pack("<B3BI", 0x17, 0x00, 0x00, 0x00, 8)


def _dt_vlen_of_objref() -> bytes:
    return struct.pack("<B3BI", 0x19, 0x00, 0x00, 0x00, 16) + _dt_objref()


def _dt_reference_list() -> bytes:
    """compound { object reference "dataset" @0; int32 "dimension" @8 } of 16 bytes (H5DS's ds_list_t), version 1."""
    def member(name: bytes, offset: int, typ: bytes) -> bytes:
        return _pad8(name + b"\0") + struct.pack("<IB3xII4I", offset, 0, 0, 0, 0, 0, 0, 0) + typ
    return (struct.pack("<B3BI", 0x16, 2, 0, 0, 16) + member(b"dataset", 0, _dt_objref())
            + member(b"dimension", 8, _dt_int32()))


def _dataspace(shape) -> bytes:
    shape = tuple(int(n) for n in shape)
    return struct.pack("<BBBBI", 1, len(shape), 0, 0, 0) + b"".join(struct.pack("<Q", n) for n in shape)


def _attribute(name: str, dtype: bytes, shape, data: bytes) -> bytes:
    nm = name.encode() + b"\0"
    ds = _dataspace(shape)
    return (struct.pack("<BBHHH", 1, 0, len(nm), len(dtype), len(ds)) + _pad8(nm) + _pad8(dtype) + _pad8(ds) + data)


def _attr_string(name: str, value: str) -> bytes:
    v = value.encode() + b"\0"
    return _attribute(name, _dt_string(len(v)), (), v)


def _attr_numeric(name: str, value) -> bytes:
    a = np.atleast_1d(np.asarray(value))
    if a.dtype.kind in "iub":
        return _attribute(name, _dt_int32(), a.shape, a.astype("<i4").tobytes())
    if a.dtype == np.float32:
        return _attribute(name, _dt_float(4), a.shape, a.astype("<f4").tobytes())
    return _attribute(name, _dt_float(8), a.shape, a.astype("<f8").tobytes())


def _object_header(messages) -> bytes:
    """version-1 object header holding `messages` = [(type, body)]."""
    body = b"".join(struct.pack("<HHB3x", t, len(_pad8(m)), 0) + _pad8(m) for t, m in messages)
    return struct.pack("<BBHII", 1, 0, len(messages), 1, len(body)) + b"\0" * 4 + body


MSG_DATASPACE, MSG_DATATYPE, MSG_FILL, MSG_LAYOUT, MSG_ATTRIBUTE, MSG_SYMTAB = 0x0001, 0x0003, 0x0005, 0x0008, 0x000C, 0x0011


class _Dataset:
    def __init__(self, name, shape, np_dtype, data):
        self.name, self.shape, self.data = name, tuple(shape), data
        self.np_dtype = np.dtype(np_dtype)
        self.attrs = []                     # callables(address_of) -> attribute message body
        self.addr = 0
        self.data_addr = UNDEF

    def dtype_msg(self) -> bytes:
        if self.np_dtype.kind == "i":
            return _dt_int32()
        return _dt_float(self.np_dtype.itemsize)

    def nbytes(self) -> int:
        return int(np.prod(self.shape, dtype=np.int64)) * self.np_dtype.itemsize if self.shape else self.np_dtype.itemsize

    def header(self, ctx) -> bytes:
        msgs = [(MSG_DATASPACE, _dataspace(self.shape)), (MSG_DATATYPE, self.dtype_msg()),
                (MSG_FILL, struct.pack("<BBBBI", 2, 2, 2, 1, 0)),
                (MSG_LAYOUT, struct.pack("<BBQQ", 3, 1, self.data_addr, self.nbytes()))]
        msgs += [(MSG_ATTRIBUTE, a(ctx)) for a in self.attrs]
        return _object_header(msgs)


def write_nc4(path: str, dims: dict, variables: dict, attrs: dict | None = None, double: bool = True) -> None:
    """Write a netCDF-4 file; same calling convention as ncfile.write_nc:
    ``variables[name] = (dim_names, array[, var_attrs])``, floating-point data as float64 if ``double`` else float32."""
    dim_names = list(dims)
    dimid = {d: i for i, d in enumerate(dim_names)}
    objs: list[_Dataset] = []
    by_name = {}
    for name, spec in variables.items():
        dn, arr = tuple(spec[0]), np.asarray(spec[1])
        vattrs = spec[2] if len(spec) > 2 else {}
        shape0 = arr.shape      # (ascontiguousarray makes a scalar one-dimensional)
        if arr.dtype.kind == "f":
            arr = np.ascontiguousarray(arr, dtype="<f8" if double else "<f4").reshape(shape0)
        else:
            arr = np.ascontiguousarray(arr, dtype="<i4").reshape(shape0)
        if tuple(arr.shape) != tuple(dims[d] for d in dn):
            raise ValueError(f"{name}: shape {arr.shape} does not match dimensions {dn}")
        ds = _Dataset(name, arr.shape, arr.dtype, arr.tobytes())
        ds.dim_names = dn
        ds.user_attrs = vattrs
        objs.append(ds)
        by_name[name] = ds
    # a dimension without a variable of its name becomes a float32 dataset that is never written (netCDF-4's own habit)
    users = {d: [] for d in dim_names}          # (variable, index of the dimension in it)
    for v in list(objs):
        for k, d in enumerate(v.dim_names):
            users[d].append((v, k))
    for d in dim_names:
        if d in by_name and by_name[d].dim_names == (d,):
            scale = by_name[d]                   # coordinate variable
            scale.is_coordinate = True
        else:
            if d in by_name:
                raise ValueError(f"variable {d} has the name of a dimension but is not its coordinate variable")
            scale = _Dataset(d, (dims[d],), "<f4", b"")
            scale.dim_names = ()
            scale.user_attrs = {}
            scale.is_coordinate = False
            objs.append(scale)
            by_name[d] = scale
        scale.is_scale = True
    objs.sort(key=lambda o: o.name.encode())      # symbol-table entries are kept in strcmp order
    if len(objs) > 2 * _LEAF_K:
        raise ValueError("too many variables for one symbol-table node")

    # global heap: one object (a sequence of one object reference) per (variable, dimension)
    gheap_items = []                              # (variable, k, dimension name)
    for v in objs:
        if not getattr(v, "is_scale", False) or getattr(v, "is_coordinate", False):
            for k, d in enumerate(v.dim_names):
                if by_name[d] is v:
                    continue                      # a coordinate variable is not attached to itself
                gheap_items.append((v, k, d))
    gheap_index = {(id(v), k): i + 1 for i, (v, k, d) in enumerate(gheap_items)}
    gheap_used = 16 + 24 * len(gheap_items)
    gheap_size = max(4096, -(-(gheap_used + 16) // 4096) * 4096)

    class Ctx:
        gheap_addr = 0

    ctx = Ctx()
    # attribute builders (they need object addresses, so they are evaluated at layout time)
    for v in objs:
        is_scale = getattr(v, "is_scale", False)
        if is_scale:
            v.attrs.append(lambda c: _attr_string("CLASS", "DIMENSION_SCALE"))
            if getattr(v, "is_coordinate", False):
                v.attrs.append(lambda c, n=v.name: _attr_string("NAME", n))
            else:
                v.attrs.append(lambda c, n=v.shape[0]: _attr_string("NAME", "%s%10d" % (NC_DIM_WITHOUT_VARIABLE, n)))
            v.attrs.append(lambda c, i=dimid[v.name]: _attribute("_Netcdf4Dimid", _dt_int32(), (), struct.pack("<i", i)))
            refs = [(u, k) for (u, k) in users[v.name] if u is not v]
            if refs:
                v.attrs.append(lambda c, refs=refs: _attribute(
                    "REFERENCE_LIST", _dt_reference_list(), (len(refs),),
                    b"".join(struct.pack("<Qi4x", u.addr, k) for u, k in refs)))
        attached = [(k, d) for k, d in enumerate(v.dim_names) if by_name[d] is not v]
        if attached:
            v.attrs.append(lambda c, v=v: _attribute(
                "DIMENSION_LIST", _dt_vlen_of_objref(), (len(v.dim_names),),
                b"".join(struct.pack("<IQI", 1, c.gheap_addr, gheap_index[(id(v), k)]) for k in range(len(v.dim_names)))))
        if v.dim_names and not (is_scale and not getattr(v, "is_coordinate", False)):
            v.attrs.append(lambda c, v=v: _attribute("_Netcdf4Coordinates", _dt_int32(), (len(v.dim_names),),
                                                      np.array([dimid[d] for d in v.dim_names], "<i4").tobytes()))
        for ak, av in v.user_attrs.items():
            v.attrs.append((lambda c, ak=ak, av=av: _attr_string(ak, av)) if isinstance(av, str)
                           else (lambda c, ak=ak, av=av: _attr_numeric(ak, av)))

    root_attrs = [_attr_string("_NCProperties", "version=2,ecrad_amd=1")]
    for k, a in (attrs or {}).items():
        root_attrs.append(_attr_string(k, a) if isinstance(a, str) else _attr_numeric(k, a))

    # ---- layout -----------------------------------------------------------------------------------------------------
    heap_data = bytearray(b"\0" * 8)               # offset 0: the empty name
    name_off = {}
    for o in objs:
        name_off[o.name] = len(heap_data)
        heap_data += _pad8(o.name.encode() + b"\0")
    heap_free = len(heap_data)
    heap_data += struct.pack("<QQ", 1, 32) + b"\0" * 16       # one free block (next = 1: end of list), 32 bytes
    btree_size = 24 + (2 * _INTERNAL_K + 1) * 8 + 2 * _INTERNAL_K * 8
    snod_size = 8 + 2 * _LEAF_K * 40

    def root_header(btree_addr, heap_addr):
        return _object_header([(MSG_SYMTAB, struct.pack("<QQ", btree_addr, heap_addr))]
                              + [(MSG_ATTRIBUTE, a) for a in root_attrs])

    pos = 96
    root_addr = pos
    pos += len(root_header(0, 0))
    btree_addr = pos
    pos += btree_size
    heap_addr = pos
    pos += 32
    heap_data_addr = pos
    pos += len(heap_data)
    snod_addr = pos
    pos += snod_size
    ctx.gheap_addr = pos
    pos += gheap_size
    for o in objs:                                 # object headers (their sizes do not depend on the addresses)
        o.addr = pos
        pos += len(o.header(ctx))
    for o in objs:
        if o.data:
            pos += -pos % 8
            o.data_addr = pos
            pos += len(o.data)
    eof = pos

    out = bytearray(eof)
    sb = (_SIG + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, _LEAF_K, _INTERNAL_K, 0)
          + struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
          + struct.pack("<QQII", 0, root_addr, 1, 0) + struct.pack("<QQ", btree_addr, heap_addr))
    assert len(sb) == 96
    out[0:96] = sb
    rh = root_header(btree_addr, heap_addr)
    out[root_addr:root_addr + len(rh)] = rh
    bt = b"TREE" + struct.pack("<BBHQQ", 0, 0, 1, UNDEF, UNDEF) + struct.pack("<QQQ", 0, snod_addr, name_off[objs[-1].name])
    out[btree_addr:btree_addr + len(bt)] = bt
    out[heap_addr:heap_addr + 32] = b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap_data), heap_free, heap_data_addr)
    out[heap_data_addr:heap_data_addr + len(heap_data)] = heap_data
    sn = b"SNOD" + struct.pack("<BBH", 1, 0, len(objs)) + b"".join(
        struct.pack("<QQII16x", name_off[o.name], o.addr, 0, 0) for o in objs)
    out[snod_addr:snod_addr + len(sn)] = sn
    gh = bytearray(b"GCOL" + struct.pack("<B3xQ", 1, gheap_size))
    for i, (v, k, d) in enumerate(gheap_items):
        gh += struct.pack("<HH4xQ", i + 1, 1, 8) + struct.pack("<Q", by_name[d].addr)
    gh += struct.pack("<HH4xQ", 0, 0, gheap_size - len(gh))            # object 0: the free space (size includes this header)
    out[ctx.gheap_addr:ctx.gheap_addr + len(gh)] = gh
    for o in objs:
        h = o.header(ctx)
        out[o.addr:o.addr + len(h)] = h
        if o.data:
            out[o.data_addr:o.data_addr + len(o.data)] = o.data
    with open(path, "wb") as f:
        f.write(out)
