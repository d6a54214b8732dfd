"""Minimal HDF5 reader/writer on the C library through ctypes.

h5py is not installed in this image, but libhdf5 is.  This module implements exactly
the subset of the h5py API the harness uses for the reference's file formats
(datasets.py:1-70, results.py:1-48): ``File(path, mode)``, ``f[name] = array``,
``f[name][:]``, ``f.attrs[key] = scalar | str | bool``, ``dict(f.attrs)``, ``close()``.
Files written here are ordinary HDF5: float64/int64 datasets, variable-length UTF-8
string attributes (what h5py writes for ``str``), the int8 enum {FALSE = 0, TRUE = 1} h5py
uses for booleans -- h5py (and so the reference's ``plot.py`` / ``create_website.py``) reads
them back unchanged.  ``tests/test_file_formats.py`` checks the files with ``h5dump -H``.
"""
import ctypes
import ctypes.util
import os

import numpy as np

_CANDIDATES = ["libhdf5.so", "libhdf5.so.103", "libhdf5_serial.so", "libhdf5_serial.so.103",
               "/opt/conda/lib/libhdf5.so", "/opt/conda/lib/libhdf5.so.103"]
_lib = None
_ids = {}

hid_t = ctypes.c_int64
hsize_t = ctypes.c_uint64
H5F_ACC_RDONLY, H5F_ACC_TRUNC = 0, 2
H5P_DEFAULT = 0
H5S_ALL = 0
H5T_INTEGER, H5T_FLOAT, H5T_STRING, H5T_ENUM = 0, 1, 3, 8
H5T_VARIABLE = ctypes.c_size_t(-1).value
H5T_CSET_UTF8 = 1
H5S_SCALAR = 0


def _load():
    global _lib
    if _lib is not None:
        return _lib
    names = list(_CANDIDATES)
    found = ctypes.util.find_library("hdf5")
    if found:
        names.insert(0, found)
    for n in names:
        try:
            lib = ctypes.CDLL(n)
        except OSError:
            continue
        if hasattr(lib, "H5Fcreate") and hasattr(lib, "H5Dcreate2"):
            _lib = lib
            break
    if _lib is None:
        return None
    L = _lib
    L.H5open.restype = ctypes.c_int
    L.H5open()
    for sym in ("H5T_NATIVE_DOUBLE_g", "H5T_NATIVE_INT64_g", "H5T_NATIVE_INT8_g", "H5T_C_S1_g",
                "H5T_NATIVE_FLOAT_g"):
        _ids[sym] = hid_t.in_dll(L, sym).value
    proto = {
        "H5Fcreate": (hid_t, [ctypes.c_char_p, ctypes.c_uint, hid_t, hid_t]),
        "H5Fopen": (hid_t, [ctypes.c_char_p, ctypes.c_uint, hid_t]),
        "H5Fclose": (ctypes.c_int, [hid_t]),
        "H5Screate_simple": (hid_t, [ctypes.c_int, ctypes.POINTER(hsize_t), ctypes.POINTER(hsize_t)]),
        "H5Screate": (hid_t, [ctypes.c_int]),
        "H5Sclose": (ctypes.c_int, [hid_t]),
        "H5Sget_simple_extent_ndims": (ctypes.c_int, [hid_t]),
        "H5Sget_simple_extent_dims": (ctypes.c_int, [hid_t, ctypes.POINTER(hsize_t), ctypes.POINTER(hsize_t)]),
        "H5Dcreate2": (hid_t, [hid_t, ctypes.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
        "H5Dopen2": (hid_t, [hid_t, ctypes.c_char_p, hid_t]),
        "H5Dwrite": (ctypes.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, ctypes.c_void_p]),
        "H5Dread": (ctypes.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, ctypes.c_void_p]),
        "H5Dget_space": (hid_t, [hid_t]),
        "H5Dget_type": (hid_t, [hid_t]),
        "H5Dclose": (ctypes.c_int, [hid_t]),
        "H5Acreate2": (hid_t, [hid_t, ctypes.c_char_p, hid_t, hid_t, hid_t, hid_t]),
        "H5Awrite": (ctypes.c_int, [hid_t, hid_t, ctypes.c_void_p]),
        "H5Aread": (ctypes.c_int, [hid_t, hid_t, ctypes.c_void_p]),
        "H5Aclose": (ctypes.c_int, [hid_t]),
        "H5Aget_num_attrs": (ctypes.c_int, [hid_t]),
        "H5Aopen_by_idx": (hid_t, [hid_t, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, hsize_t, hid_t, hid_t]),
        "H5Aget_name": (ctypes.c_ssize_t, [hid_t, ctypes.c_size_t, ctypes.c_char_p]),
        "H5Aget_type": (hid_t, [hid_t]),
        "H5Aget_space": (hid_t, [hid_t]),
        "H5Adelete": (ctypes.c_int, [hid_t, ctypes.c_char_p]),
        "H5Aexists": (ctypes.c_int, [hid_t, ctypes.c_char_p]),
        "H5Tcopy": (hid_t, [hid_t]),
        "H5Tenum_create": (hid_t, [hid_t]),
        "H5Tenum_insert": (ctypes.c_int, [hid_t, ctypes.c_char_p, ctypes.c_void_p]),
        "H5Tset_size": (ctypes.c_int, [hid_t, ctypes.c_size_t]),
        "H5Tset_cset": (ctypes.c_int, [hid_t, ctypes.c_int]),
        "H5Tget_class": (ctypes.c_int, [hid_t]),
        "H5Tget_size": (ctypes.c_size_t, [hid_t]),
        "H5Tis_variable_str": (ctypes.c_int, [hid_t]),
        "H5Tclose": (ctypes.c_int, [hid_t]),
        "H5Lexists": (ctypes.c_int, [hid_t, ctypes.c_char_p, hid_t]),
        "H5Literate": None,
        "H5Gget_num_objs": (ctypes.c_int, [hid_t, ctypes.POINTER(hsize_t)]),
        "H5Gget_objname_by_idx": (ctypes.c_ssize_t, [hid_t, hsize_t, ctypes.c_char_p, ctypes.c_size_t]),
        "H5Eset_auto2": (ctypes.c_int, [hid_t, ctypes.c_void_p, ctypes.c_void_p]),
        "H5Gopen2": (hid_t, [hid_t, ctypes.c_char_p, hid_t]),
        "H5Gclose": (ctypes.c_int, [hid_t]),
    }
    for name, sig in proto.items():
        if sig is None or not hasattr(L, name):
            continue
        fn = getattr(L, name)
        fn.restype, fn.argtypes = sig
    L.H5Eset_auto2(0, None, None)  # errors are reported through return codes -> exceptions
    return _lib


def available():
    return _load() is not None


def _check(v, what):
    if v < 0:
        raise OSError(f"HDF5 call failed: {what}")
    return v


class _AttrView:
    """dict-like view of the attributes of the root group."""

    def __init__(self, file):
        self._f = file

    def __setitem__(self, key, value):
        L, fid = _load(), self._f._root
        k = key.encode()
        if L.H5Aexists(fid, k) > 0:
            L.H5Adelete(fid, k)
        space = _check(L.H5Screate(H5S_SCALAR), "H5Screate")
        try:
            if isinstance(value, bytes):
                value = value.decode()
            if isinstance(value, str):
                t = _check(L.H5Tcopy(_ids["H5T_C_S1_g"]), "H5Tcopy")
                L.H5Tset_size(t, H5T_VARIABLE)
                L.H5Tset_cset(t, H5T_CSET_UTF8)
                buf = ctypes.c_char_p(value.encode("utf-8"))
                a = _check(L.H5Acreate2(fid, k, t, space, H5P_DEFAULT, H5P_DEFAULT), "H5Acreate2")
                _check(L.H5Awrite(a, t, ctypes.byref(buf)), "H5Awrite")
                L.H5Aclose(a)
                L.H5Tclose(t)
                return
            if isinstance(value, (bool, np.bool_)):
                # numpy.bool_ in h5py: an enum over int8 with members FALSE = 0, TRUE = 1
                t = _check(L.H5Tenum_create(_ids["H5T_NATIVE_INT8_g"]), "H5Tenum_create")
                for member, code in ((b"FALSE", 0), (b"TRUE", 1)):
                    c8 = ctypes.c_int8(code)
                    _check(L.H5Tenum_insert(t, member, ctypes.byref(c8)), "H5Tenum_insert")
                v = ctypes.c_int8(int(bool(value)))
                a = _check(L.H5Acreate2(fid, k, t, space, H5P_DEFAULT, H5P_DEFAULT), "H5Acreate2")
                _check(L.H5Awrite(a, t, ctypes.byref(v)), "H5Awrite")
                L.H5Aclose(a)
                L.H5Tclose(t)
                return
            if isinstance(value, (int, np.integer)):
                t, v = _ids["H5T_NATIVE_INT64_g"], ctypes.c_int64(int(value))
            elif isinstance(value, (float, np.floating)):
                t, v = _ids["H5T_NATIVE_DOUBLE_g"], ctypes.c_double(float(value))
            else:
                raise TypeError(f"attribute {key!r}: unsupported type {type(value)}")
            a = _check(L.H5Acreate2(fid, k, t, space, H5P_DEFAULT, H5P_DEFAULT), "H5Acreate2")
            _check(L.H5Awrite(a, t, ctypes.byref(v)), "H5Awrite")
            L.H5Aclose(a)
        finally:
            L.H5Sclose(space)

    def _read_all(self):
        L, fid = _load(), self._f._root
        out = {}
        n = _check(L.H5Aget_num_attrs(fid), "H5Aget_num_attrs")
        for i in range(n):
            a = _check(L.H5Aopen_by_idx(fid, b".", 0, 0, i, H5P_DEFAULT, H5P_DEFAULT), "H5Aopen_by_idx")
            size = L.H5Aget_name(a, 0, None)
            nb = ctypes.create_string_buffer(size + 1)
            L.H5Aget_name(a, size + 1, nb)
            name = nb.value.decode()
            t = L.H5Aget_type(a)
            cls = L.H5Tget_class(t)
            if cls == H5T_STRING:
                if L.H5Tis_variable_str(t) > 0:
                    p = ctypes.c_char_p()
                    mt = L.H5Tcopy(_ids["H5T_C_S1_g"])
                    L.H5Tset_size(mt, H5T_VARIABLE)
                    L.H5Tset_cset(mt, H5T_CSET_UTF8)
                    _check(L.H5Aread(a, mt, ctypes.byref(p)), "H5Aread")
                    out[name] = (p.value or b"").decode("utf-8")
                    L.H5Tclose(mt)
                else:
                    sz = L.H5Tget_size(t)
                    buf = ctypes.create_string_buffer(sz + 1)
                    _check(L.H5Aread(a, t, buf), "H5Aread")
                    out[name] = buf.value.decode("utf-8")
            elif cls == H5T_FLOAT:
                v = ctypes.c_double()
                _check(L.H5Aread(a, _ids["H5T_NATIVE_DOUBLE_g"], ctypes.byref(v)), "H5Aread")
                out[name] = v.value
            elif cls in (H5T_INTEGER, H5T_ENUM):
                v = ctypes.c_int64()
                if cls == H5T_ENUM:  # h5py booleans: enum over int8
                    v8 = ctypes.c_int8()
                    _check(L.H5Aread(a, t, ctypes.byref(v8)), "H5Aread")
                    out[name] = bool(v8.value)
                else:
                    sz = L.H5Tget_size(t)
                    _check(L.H5Aread(a, _ids["H5T_NATIVE_INT64_g"], ctypes.byref(v)), "H5Aread")
                    out[name] = bool(v.value) if sz == 1 else int(v.value)
            L.H5Tclose(t)
            L.H5Aclose(a)
        return out

    def __getitem__(self, key):
        return self._read_all()[key]

    def __contains__(self, key):
        return _load().H5Aexists(self._f._root, key.encode()) > 0

    def get(self, key, default=None):
        return self._read_all().get(key, default)

    def keys(self):
        return self._read_all().keys()

    def items(self):
        return self._read_all().items()

    def __iter__(self):
        return iter(self._read_all())

    def __len__(self):
        return len(self._read_all())


class File:
    def __init__(self, path, mode="r"):
        L = _load()
        if L is None:
            raise OSError("libhdf5 not found")
        self.path, self.mode = path, mode
        if mode == "r":
            self._fid = _check(L.H5Fopen(path.encode(), H5F_ACC_RDONLY, H5P_DEFAULT), f"open {path}")
        elif mode == "w":
            self._fid = _check(L.H5Fcreate(path.encode(), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT), f"create {path}")
        else:
            raise ValueError("mode must be 'r' or 'w'")
        self._root = _check(L.H5Gopen2(self._fid, b"/", H5P_DEFAULT), "open root group")
        self.attrs = _AttrView(self)

    def __setitem__(self, name, value):
        L = _load()
        arr = np.asarray(value)
        if arr.dtype.kind in "iub":
            arr = np.ascontiguousarray(arr, dtype=np.int64)
            t = _ids["H5T_NATIVE_INT64_g"]
        else:
            arr = np.ascontiguousarray(arr, dtype=np.float64)
            t = _ids["H5T_NATIVE_DOUBLE_g"]
        dims = (hsize_t * max(arr.ndim, 1))(*arr.shape) if arr.ndim else None
        space = _check(L.H5Screate_simple(arr.ndim, dims, None) if arr.ndim else L.H5Screate(H5S_SCALAR), "H5Screate")
        d = _check(L.H5Dcreate2(self._fid, name.encode(), t, space, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), f"create dataset {name}")
        if arr.size:
            _check(L.H5Dwrite(d, t, H5S_ALL, H5S_ALL, H5P_DEFAULT, arr.ctypes.data), f"write {name}")
        L.H5Dclose(d)
        L.H5Sclose(space)

    def __contains__(self, name):
        return _load().H5Lexists(self._fid, name.encode(), H5P_DEFAULT) > 0

    def __getitem__(self, name):
        L = _load()
        d = L.H5Dopen2(self._fid, name.encode(), H5P_DEFAULT)
        if d < 0:
            raise KeyError(name)
        try:
            space = L.H5Dget_space(d)
            nd = L.H5Sget_simple_extent_ndims(space)
            dims = (hsize_t * max(nd, 1))()
            if nd:
                L.H5Sget_simple_extent_dims(space, dims, None)
            shape = tuple(int(dims[i]) for i in range(nd))
            t = L.H5Dget_type(d)
            cls = L.H5Tget_class(t)
            L.H5Tclose(t)
            L.H5Sclose(space)
            if cls == H5T_FLOAT:
                out, mt = np.empty(shape, dtype=np.float64), _ids["H5T_NATIVE_DOUBLE_g"]
            elif cls == H5T_INTEGER:
                out, mt = np.empty(shape, dtype=np.int64), _ids["H5T_NATIVE_INT64_g"]
            else:
                raise TypeError(f"dataset {name}: unsupported type class {cls}")
            if out.size:
                _check(L.H5Dread(d, mt, H5S_ALL, H5S_ALL, H5P_DEFAULT, out.ctypes.data), f"read {name}")
            return out
        finally:
            L.H5Dclose(d)

    def keys(self):
        L = _load()
        n = hsize_t()
        L.H5Gget_num_objs(self._root, ctypes.byref(n))
        names = []
        for i in range(n.value):
            size = L.H5Gget_objname_by_idx(self._root, i, None, 0)
            buf = ctypes.create_string_buffer(size + 1)
            L.H5Gget_objname_by_idx(self._root, i, buf, size + 1)
            names.append(buf.value.decode())
        return names

    def close(self):
        if getattr(self, "_fid", None) is not None and self._fid >= 0:
            if getattr(self, "_root", -1) >= 0:
                _load().H5Gclose(self._root)
                self._root = -1
            _load().H5Fclose(self._fid)
            self._fid = -1

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
