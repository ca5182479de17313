"""Reading and writing the reference's checkpoint container (HDF5) without h5py.

The reference saves its weights with ``chainer.serializers.save_hdf5`` (wavenet.py:619-625).  Chainer's ``HDF5Serializer``
writes one group per link and one dataset per parameter (``/<link name>/W``, ``/<link name>/b``), gzip level 4 for every
array with more than one element.  h5py is not installed for this image's interpreter, but the HDF5 C library that h5py
itself wraps is on the image (``libhdf5.so``); this module binds the dozen entry points it needs through ctypes.

    read_datasets(path)            -> {"causal_0/W": ndarray, ...}   every dataset of the file, keyed by its path
    write_datasets(path, arrays)   the same layout Chainer writes (groups from the "/" in the keys, gzip 4)

The library is looked for in $WAVENET_HDF5_LIB, then by the loader's own search, then in the usual prefixes; `available()`
says whether one was found, and both functions raise ImportError naming what was tried when none was.
"""
from __future__ import annotations

import ctypes as C
import ctypes.util
import glob
import os
from typing import Dict, Optional

import numpy as np

_hid = C.c_int64                 # hid_t since HDF5 1.10
_hsize = C.c_uint64
_H5F_ACC_RDONLY, _H5F_ACC_TRUNC = 0, 2
_H5I_GROUP, _H5I_DATASET = 2, 5
_H5T_INTEGER, _H5T_FLOAT = 0, 1
_lib: Optional[C.CDLL] = None
_tried: list = []


def _candidates():
    env = os.environ.get("WAVENET_HDF5_LIB")
    if env:
        yield env
    found = ctypes.util.find_library("hdf5")
    if found:
        yield found
    for pat in ("/usr/lib/x86_64-linux-gnu/libhdf5*.so*", "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so*",
                "/usr/local/lib/libhdf5.so*", "/opt/conda/lib/libhdf5.so*"):
        for p in sorted(glob.glob(pat)):
            if "_hl" not in p and "_cpp" not in p and "fortran" not in p:
                yield p


def _load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    for p in _candidates():
        _tried.append(p)
        try:
            lib = C.CDLL(p)
            major, minor, rel = C.c_uint(), C.c_uint(), C.c_uint()
            if lib.H5open() < 0 or lib.H5get_libversion(C.byref(major), C.byref(minor), C.byref(rel)) < 0:
                continue
            if (major.value, minor.value) < (1, 10):      # hid_t is 32 bits before 1.10
                continue
        except (OSError, AttributeError):
            continue
        sig = {
            "H5Fopen": (_hid, [C.c_char_p, C.c_uint, _hid]), "H5Fcreate": (_hid, [C.c_char_p, C.c_uint, _hid, _hid]),
            "H5Fclose": (C.c_int, [_hid]), "H5Oopen": (_hid, [_hid, C.c_char_p, _hid]), "H5Oclose": (C.c_int, [_hid]),
            "H5Iget_type": (C.c_int, [_hid]), "H5Dget_space": (_hid, [_hid]), "H5Dget_type": (_hid, [_hid]),
            "H5Sget_simple_extent_ndims": (C.c_int, [_hid]),
            "H5Sget_simple_extent_dims": (C.c_int, [_hid, C.POINTER(_hsize), C.POINTER(_hsize)]),
            "H5Sclose": (C.c_int, [_hid]), "H5Tget_class": (C.c_int, [_hid]), "H5Tget_size": (C.c_size_t, [_hid]),
            "H5Tclose": (C.c_int, [_hid]), "H5Dread": (C.c_int, [_hid, _hid, _hid, _hid, _hid, C.c_void_p]),
            "H5Dwrite": (C.c_int, [_hid, _hid, _hid, _hid, _hid, C.c_void_p]),
            "H5Dcreate2": (_hid, [_hid, C.c_char_p, _hid, _hid, _hid, _hid, _hid]), "H5Dclose": (C.c_int, [_hid]),
            "H5Gcreate2": (_hid, [_hid, C.c_char_p, _hid, _hid, _hid]), "H5Gclose": (C.c_int, [_hid]),
            "H5Lexists": (C.c_int, [_hid, C.c_char_p, _hid]),
            "H5Screate_simple": (_hid, [C.c_int, C.POINTER(_hsize), C.POINTER(_hsize)]), "H5Screate": (_hid, [C.c_int]),
            "H5Pcreate": (_hid, [_hid]), "H5Pclose": (C.c_int, [_hid]),
            "H5Pset_chunk": (C.c_int, [_hid, C.c_int, C.POINTER(_hsize)]), "H5Pset_deflate": (C.c_int, [_hid, C.c_uint]),
            "H5Eset_auto2": (C.c_int, [_hid, C.c_void_p, C.c_void_p]),
        }
        # H5Lvisit is a plain symbol in 1.10 only; from 1.12 on it is a versioned macro and the library exports H5Lvisit2
        # (and H5Lvisit1).  The callback ignores its info argument, so one prototype serves all three.
        visit = None
        for vname in ("H5Lvisit2", "H5Lvisit1", "H5Lvisit"):
            if hasattr(lib, vname):
                visit = getattr(lib, vname)
                break
        if visit is None:
            continue
        try:
            for name, (res, args) in sig.items():
                fn = getattr(lib, name)
                fn.restype, fn.argtypes = res, args
        except AttributeError:
            continue
        visit.restype, visit.argtypes = C.c_int, [_hid, C.c_int, C.c_int, _LINK_CB, C.c_void_p]
        lib._wn_lvisit = visit
        lib.H5Eset_auto2(0, None, None)                    # errors come back as return codes, not as a printed stack
        _lib = lib
        return lib
    raise ImportError("no HDF5 C library (>= 1.10) found for reading / writing the reference's checkpoints; tried %s. "
                      "Set WAVENET_HDF5_LIB to a libhdf5.so, or install h5py." % (_tried or ["nothing"]))


def available() -> bool:
    try:
        _load()
        return True
    except ImportError:
        return False


def _g(lib, name) -> int:
    """Value of one of the library's hid_t globals (H5T_NATIVE_FLOAT and friends are macros over these)."""
    return _hid.in_dll(lib, name).value


_LINK_CB = C.CFUNCTYPE(C.c_int, _hid, C.c_char_p, C.c_void_p, C.c_void_p)


def read_datasets(path: str) -> Dict[str, np.ndarray]:
    """Every dataset of the file, keyed by its path without the leading "/".  Floating-point data come back in their
    stored width (float32 / float64), integers as int64; compressed (gzip) datasets are inflated by the library."""
    lib = _load()
    f = lib.H5Fopen(os.fsencode(path), _H5F_ACC_RDONLY, 0)
    if f < 0:
        raise OSError("cannot open %s as HDF5" % path)
    names: list = []
    cb = _LINK_CB(lambda g, name, info, data: names.append(name) or 0)
    out: Dict[str, np.ndarray] = {}
    try:
        if lib._wn_lvisit(f, 0, 0, cb, None) < 0:           # H5_INDEX_NAME, H5_ITER_INC
            raise OSError("walking the links of %s failed" % path)
        for name in names:
            o = lib.H5Oopen(f, name, 0)
            if o < 0:
                continue
            try:
                if lib.H5Iget_type(o) != _H5I_DATASET:
                    continue
                sp, ty = lib.H5Dget_space(o), lib.H5Dget_type(o)
                nd = lib.H5Sget_simple_extent_ndims(sp)
                dims = (_hsize * max(nd, 1))()
                if nd > 0:
                    lib.H5Sget_simple_extent_dims(sp, dims, None)
                shape = tuple(int(dims[i]) for i in range(nd))
                cls, size = lib.H5Tget_class(ty), lib.H5Tget_size(ty)
                lib.H5Sclose(sp); lib.H5Tclose(ty)
                if cls == _H5T_FLOAT:
                    dt, mem = (np.float64, "H5T_NATIVE_DOUBLE_g") if size == 8 else (np.float32, "H5T_NATIVE_FLOAT_g")
                elif cls == _H5T_INTEGER:
                    dt, mem = np.int64, "H5T_NATIVE_INT64_g"
                else:
                    continue                                   # strings, compounds: nothing Chainer writes for a model
                a = np.empty(shape, dtype=dt)
                if lib.H5Dread(o, _g(lib, mem), 0, 0, 0, a.ctypes.data_as(C.c_void_p)) < 0:
                    raise OSError("reading dataset %s of %s failed (a filter the library was built without?)"
                                  % (name.decode(), path))
                out[name.decode()] = a
            finally:
                lib.H5Oclose(o)
    finally:
        lib.H5Fclose(f)
    return out


def write_datasets(path: str, arrays: Dict[str, np.ndarray], compression: int = 4) -> None:
    """One dataset per key, one group per "/" in it -- what chainer.serializers.save_hdf5 produces for a Chain of links:
    float32 data stay float32, other floats go to float64, integers to int64; gzip `compression` for arrays with more than
    one element (HDF5Serializer's rule), contiguous otherwise."""
    lib = _load()
    f = lib.H5Fcreate(os.fsencode(path), _H5F_ACC_TRUNC, 0, 0)
    if f < 0:
        raise OSError("cannot create %s" % path)
    try:
        for key in sorted(arrays):
            a = np.asarray(arrays[key])
            if a.dtype == np.float32:
                mem, filet = "H5T_NATIVE_FLOAT_g", "H5T_IEEE_F32LE_g"
            elif a.dtype.kind == "f":
                a, mem, filet = a.astype(np.float64), "H5T_NATIVE_DOUBLE_g", "H5T_IEEE_F64LE_g"
            elif a.dtype.kind in "iub":
                a, mem, filet = a.astype(np.int64), "H5T_NATIVE_INT64_g", "H5T_STD_I64LE_g"
            else:
                raise TypeError("%s: dtype %s has no HDF5 mapping here" % (key, a.dtype))
            a = np.ascontiguousarray(a).reshape(a.shape)       # (ascontiguousarray turns a 0-d array into shape (1,))
            parts = key.strip("/").split("/")
            for i in range(1, len(parts)):                    # the groups on the way
                g = "/".join(parts[:i]).encode()
                if lib.H5Lexists(f, g, 0) <= 0:
                    gid = lib.H5Gcreate2(f, g, 0, 0, 0)
                    if gid < 0:
                        raise OSError("cannot create group %s in %s" % (g.decode(), path))
                    lib.H5Gclose(gid)
            if a.ndim:
                dims = (_hsize * a.ndim)(*a.shape)
                sp = lib.H5Screate_simple(a.ndim, dims, None)
            else:
                sp = lib.H5Screate(0)                          # H5S_SCALAR
            dcpl = lib.H5Pcreate(_g(lib, "H5P_CLS_DATASET_CREATE_ID_g"))
            if a.size > 1 and compression:
                lib.H5Pset_chunk(dcpl, a.ndim, dims)
                lib.H5Pset_deflate(dcpl, int(compression))
            d = lib.H5Dcreate2(f, "/".join(parts).encode(), _g(lib, filet), sp, 0, dcpl, 0)
            try:
                if d < 0 or lib.H5Dwrite(d, _g(lib, mem), 0, 0, 0, a.ctypes.data_as(C.c_void_p)) < 0:
                    raise OSError("writing dataset %s of %s failed" % (key, path))
            finally:
                if d >= 0:
                    lib.H5Dclose(d)
                lib.H5Pclose(dcpl); lib.H5Sclose(sp)
    finally:
        lib.H5Fclose(f)
