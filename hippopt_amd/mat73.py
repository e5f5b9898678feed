"""MAT v7.3 (= HDF5) container in the layout `hdf5storage.savemat` produces in its MATLAB-compatible mode — the format of the
reference's result files (`hdf5storage.savemat(file_name=..., mdict=..., truncate_existing=True)`,
turnkey_planners/humanoid_kinodynamic/main_periodic_step.py:509-513 and the other main scripts).

Neither hdf5storage nor h5py is in this image, but the HDF5 C library is (libhdf5 1.10, /opt/conda/lib): it is bound here with
ctypes, no Python package needed.  Where the library is missing `available()` is False and `serialization.save_mat` keeps writing
MAT v5.

Layout (the MATLAB v7.3 conventions hdf5storage follows with `matlab_compatible=True`):
  * 512-byte user block in front of the HDF5 superblock, its first 128 bytes the MAT header
    ("MATLAB 7.3 MAT-file, Platform: ..., Created on: ... HDF5 schema 1.00 ." padded to 116 bytes, 8 zero bytes, 0x0200, "IM");
  * dict -> group, attribute MATLAB_class = "struct" (+ MATLAB_fields, the field order, when every name is a valid MATLAB field name);
  * numeric ndarray -> dataset with the dimensions REVERSED (MATLAB is column major), MATLAB_class = "double" / "int64" / ...;
    one-dimensional arrays are rows (hdf5storage's `oned_as='row'`), scalars 1 x 1;
  * bool -> uint8, MATLAB_class = "logical", MATLAB_int_decode = 1;
  * str -> UTF-16 code units as uint16, MATLAB_class = "char", MATLAB_int_decode = 2;
  * list / tuple / object ndarray -> cell array: a dataset of object references (MATLAB_class = "cell") to members of the group
    "/#refs#", whose member "a" is the canonical empty every v7.3 file with cells carries;
  * empty array -> uint64 dataset holding the SHAPE, MATLAB_empty = 1;
  * arrays of at least 16 KB: one gzip(7) + shuffle + fletcher32 chunk (hdf5storage's defaults).
The Python.* attributes hdf5storage adds with `store_python_metadata=True` are NOT written: a file from here reads back in
MATLAB and through `hdf5storage.loadmat` as MATLAB types (struct -> dict, cell -> object array, row vectors 1 x n).
`loadmat` here reads both this writer's files and files with those conventions written elsewhere (filters are the library's
business), returning dict / list / str / ndarray with MATLAB's dimensions squeezed, like `serialization.load_mat` does for v5."""
import ctypes as C
import ctypes.util
import datetime
import glob
import os
import re
import sys

import numpy as np

_hid = C.c_int64
_hsize = C.c_uint64
_LIB = None
_ERR = None

H5F_ACC_RDONLY, H5F_ACC_TRUNC = 0, 2
H5S_SCALAR = 0
H5R_OBJECT = 0
H5I_GROUP, H5I_DATASET = 2, 5
H5T_INTEGER, H5T_FLOAT, H5T_STRING, H5T_REFERENCE, H5T_ENUM = 0, 1, 3, 7, 8
H5T_DIR_ASCEND = 1
H5T_STR_NULLPAD = 1
H5_INDEX_NAME, H5_ITER_INC = 0, 0


class _GInfo(C.Structure):
    _fields_ = [("storage_type", C.c_int), ("nlinks", _hsize), ("max_corder", C.c_int64), ("mounted", C.c_uint)]


class _Hvl(C.Structure):
    _fields_ = [("len", C.c_size_t), ("p", C.c_void_p)]


def _candidates():
    env = os.environ.get("HIPNLP_HDF5_LIB")
    if env:
        yield env
    found = ctypes.util.find_library("hdf5")
    if found:
        yield found
    for pat in ("/opt/conda/lib/libhdf5.so*", "/usr/lib/x86_64-linux-gnu/libhdf5*.so*", "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so*",
                "/usr/lib64/libhdf5.so*", "/usr/local/lib/libhdf5.so*", os.path.join(sys.prefix, "lib", "libhdf5.so*")):
        for p in sorted(glob.glob(pat)):
            if "_hl" not in p and "_cpp" not in p and "fortran" not in p:
                yield p


def _lib():
    global _LIB, _ERR
    if _LIB is not None or _ERR is not None:
        return _LIB
    last = None
    for path in _candidates():
        try:
            L = C.CDLL(path)
            L.H5open.restype = C.c_int
            if L.H5open() < 0:
                raise OSError("H5open failed")
            ver = (C.c_uint * 3)()
            L.H5get_libversion(C.byref(ver, 0), C.byref(ver, 4), C.byref(ver, 8))
            if (ver[0], ver[1]) < (1, 10):   # hid_t is a 32-bit int before 1.10: the 64-bit prototypes below would read garbage
                raise OSError("HDF5 %d.%d.%d is older than 1.10" % tuple(ver))
            _LIB = _Api(L, path)
            return _LIB
        except (OSError, AttributeError) as e:   # a library without one of the symbols is as good as none
            last = e
    _ERR = "no usable HDF5 C library (set HIPNLP_HDF5_LIB): %s" % (last,)
    return None


def available() -> bool:
    return _lib() is not None


class _Api:
    def __init__(self, L, path):
        self.L, self.path = L, path

        def fn(name, res, *args):
            f = getattr(L, name)
            f.restype, f.argtypes = res, list(args)
            return f

        def glob_id(name):
            return _hid.in_dll(L, name).value
        i, p, s = C.c_int, C.c_void_p, C.c_char_p
        self.Pcreate = fn("H5Pcreate", _hid, _hid)
        self.Pclose = fn("H5Pclose", i, _hid)
        self.Pset_userblock = fn("H5Pset_userblock", i, _hid, _hsize)
        self.Pset_chunk = fn("H5Pset_chunk", i, _hid, i, C.POINTER(_hsize))
        self.Pset_deflate = fn("H5Pset_deflate", i, _hid, C.c_uint)
        self.Pset_shuffle = fn("H5Pset_shuffle", i, _hid)
        self.Pset_fletcher32 = fn("H5Pset_fletcher32", i, _hid)
        self.Fcreate = fn("H5Fcreate", _hid, s, C.c_uint, _hid, _hid)
        self.Fopen = fn("H5Fopen", _hid, s, C.c_uint, _hid)
        self.Fclose = fn("H5Fclose", i, _hid)
        self.Gcreate = fn("H5Gcreate2", _hid, _hid, s, _hid, _hid, _hid)
        self.Gclose = fn("H5Gclose", i, _hid)
        self.Gget_info = fn("H5Gget_info", i, _hid, C.POINTER(_GInfo))
        self.Lexists = fn("H5Lexists", i, _hid, s, _hid)
        self.Lget_name_by_idx = fn("H5Lget_name_by_idx", C.c_ssize_t, _hid, s, i, i, _hsize, p, C.c_size_t, _hid)
        self.Oopen = fn("H5Oopen", _hid, _hid, s, _hid)
        self.Oclose = fn("H5Oclose", i, _hid)
        self.Iget_type = fn("H5Iget_type", i, _hid)
        self.Screate = fn("H5Screate", _hid, i)
        self.Screate_simple = fn("H5Screate_simple", _hid, i, C.POINTER(_hsize), C.POINTER(_hsize))
        self.Sclose = fn("H5Sclose", i, _hid)
        self.Sget_ndims = fn("H5Sget_simple_extent_ndims", i, _hid)
        self.Sget_dims = fn("H5Sget_simple_extent_dims", i, _hid, C.POINTER(_hsize), C.POINTER(_hsize))
        self.Sget_npoints = fn("H5Sget_simple_extent_npoints", C.c_int64, _hid)
        self.Dcreate = fn("H5Dcreate2", _hid, _hid, s, _hid, _hid, _hid, _hid, _hid)
        self.Dwrite = fn("H5Dwrite", i, _hid, _hid, _hid, _hid, _hid, p)
        self.Dread = fn("H5Dread", i, _hid, _hid, _hid, _hid, _hid, p)
        self.Dget_space = fn("H5Dget_space", _hid, _hid)
        self.Dget_type = fn("H5Dget_type", _hid, _hid)
        self.Dclose = fn("H5Dclose", i, _hid)
        self.Dvlen_reclaim = fn("H5Dvlen_reclaim", i, _hid, _hid, _hid, p)
        self.Acreate = fn("H5Acreate2", _hid, _hid, s, _hid, _hid, _hid, _hid)
        self.Awrite = fn("H5Awrite", i, _hid, _hid, p)
        self.Aexists = fn("H5Aexists", i, _hid, s)
        self.Aopen = fn("H5Aopen", _hid, _hid, s, _hid)
        self.Aread = fn("H5Aread", i, _hid, _hid, p)
        self.Aget_type = fn("H5Aget_type", _hid, _hid)
        self.Aget_space = fn("H5Aget_space", _hid, _hid)
        self.Aclose = fn("H5Aclose", i, _hid)
        self.Tcopy = fn("H5Tcopy", _hid, _hid)
        self.Tset_size = fn("H5Tset_size", i, _hid, C.c_size_t)
        self.Tset_strpad = fn("H5Tset_strpad", i, _hid, i)
        self.Tget_size = fn("H5Tget_size", C.c_size_t, _hid)
        self.Tget_class = fn("H5Tget_class", i, _hid)
        self.Tget_sign = fn("H5Tget_sign", i, _hid)
        self.Tget_native = fn("H5Tget_native_type", _hid, _hid, i)
        self.Tis_vstr = fn("H5Tis_variable_str", i, _hid)
        self.Tvlen_create = fn("H5Tvlen_create", _hid, _hid)
        self.Tclose = fn("H5Tclose", i, _hid)
        self.Rcreate = fn("H5Rcreate", i, p, _hid, s, i, _hid)
        self.Rdereference = fn("H5Rdereference2", _hid, _hid, _hid, i, p)
        self.Eset_auto = fn("H5Eset_auto2", i, _hid, p, p)
        self.Eset_auto(0, None, None)   # errors are reported through return codes (raised below), not printed by the library
        self.P_FILE_CREATE = glob_id("H5P_CLS_FILE_CREATE_ID_g")
        self.P_DATASET_CREATE = glob_id("H5P_CLS_DATASET_CREATE_ID_g")
        self.T = {np.dtype(k): glob_id(v) for k, v in {
            "float64": "H5T_NATIVE_DOUBLE_g", "float32": "H5T_NATIVE_FLOAT_g", "int8": "H5T_NATIVE_INT8_g", "uint8": "H5T_NATIVE_UINT8_g",
            "int16": "H5T_NATIVE_INT16_g", "uint16": "H5T_NATIVE_UINT16_g", "int32": "H5T_NATIVE_INT32_g", "uint32": "H5T_NATIVE_UINT32_g",
            "int64": "H5T_NATIVE_INT64_g", "uint64": "H5T_NATIVE_UINT64_g"}.items()}
        self.T_C_S1 = glob_id("H5T_C_S1_g")
        self.T_REF = glob_id("H5T_STD_REF_OBJ_g")


class Mat73Error(RuntimeError):
    pass


def _ck(v, what):
    if v < 0:
        raise Mat73Error("HDF5: %s failed" % what)
    return v


def _need():
    h = _lib()
    if h is None:
        raise Mat73Error(_ERR)
    return h


_FIELD_RE = re.compile(r"^[A-Za-z][A-Za-z0-9_]{0,62}$")
_MATLAB_CLASS = {np.dtype(k): k for k in ("int8", "uint8", "int16", "uint16", "int32", "uint32", "int64", "uint64")}
_MATLAB_CLASS[np.dtype("float64")] = "double"
_MATLAB_CLASS[np.dtype("float32")] = "single"
COMPRESS_THRESHOLD = 16 * 1024   # hdf5storage's compress_size_threshold


class _Writer:
    def __init__(self, h, fid, compress):
        self.h, self.fid, self.compress = h, fid, compress
        self.refs = None
        self.nref = 0

    # ---- attributes
    def attr_str(self, obj, name, value: str):
        h = self.h
        raw = value.encode("ascii")
        t = _ck(h.Tcopy(h.T_C_S1), "H5Tcopy")
        h.Tset_size(t, max(len(raw), 1))
        h.Tset_strpad(t, H5T_STR_NULLPAD)
        sp = _ck(h.Screate(H5S_SCALAR), "H5Screate")
        a = _ck(h.Acreate(obj, name.encode(), t, sp, 0, 0), "H5Acreate " + name)
        buf = C.create_string_buffer(raw, max(len(raw), 1))
        _ck(h.Awrite(a, t, buf), "H5Awrite " + name)
        h.Aclose(a); h.Sclose(sp); h.Tclose(t)

    def attr_num(self, obj, name, value, dtype):
        h = self.h
        dt = np.dtype(dtype)
        sp = _ck(h.Screate(H5S_SCALAR), "H5Screate")
        a = _ck(h.Acreate(obj, name.encode(), h.T[dt], sp, 0, 0), "H5Acreate " + name)
        v = np.array(value, dtype=dt)
        _ck(h.Awrite(a, h.T[dt], v.ctypes.data_as(C.c_void_p)), "H5Awrite " + name)
        h.Aclose(a); h.Sclose(sp)

    def attr_fields(self, obj, names):
        """MATLAB_fields: one variable-length sequence of single characters per field, in order"""
        h = self.h
        c1 = _ck(h.Tcopy(h.T_C_S1), "H5Tcopy")   # (size 1)
        vt = _ck(h.Tvlen_create(c1), "H5Tvlen_create")
        dims = (_hsize * 1)(len(names))
        sp = _ck(h.Screate_simple(1, dims, None), "H5Screate_simple")
        a = _ck(h.Acreate(obj, b"MATLAB_fields", vt, sp, 0, 0), "H5Acreate MATLAB_fields")
        keep = [C.create_string_buffer(n.encode("ascii"), len(n)) for n in names]
        arr = (_Hvl * len(names))()
        for k, b in enumerate(keep):
            arr[k].len, arr[k].p = len(names[k]), C.cast(b, C.c_void_p)
        _ck(h.Awrite(a, vt, arr), "H5Awrite MATLAB_fields")
        h.Aclose(a); h.Sclose(sp); h.Tclose(vt); h.Tclose(c1)

    # ---- datasets
    def dataset(self, parent, name, arr, h5type, matlab_class, extra=()):
        """arr: C-contiguous buffer ALREADY in file order (MATLAB dimensions reversed)"""
        h = self.h
        dims = (_hsize * arr.ndim)(*arr.shape)
        sp = _ck(h.Screate_simple(arr.ndim, dims, None), "H5Screate_simple")
        dcpl = 0
        if self.compress and arr.nbytes >= COMPRESS_THRESHOLD:
            dcpl = _ck(h.Pcreate(h.P_DATASET_CREATE), "H5Pcreate")
            _ck(h.Pset_chunk(dcpl, arr.ndim, dims), "H5Pset_chunk")
            _ck(h.Pset_shuffle(dcpl), "H5Pset_shuffle")
            _ck(h.Pset_deflate(dcpl, 7), "H5Pset_deflate")
            _ck(h.Pset_fletcher32(dcpl), "H5Pset_fletcher32")
        d = _ck(h.Dcreate(parent, name.encode("utf-8"), h5type, sp, 0, dcpl, 0), "H5Dcreate " + name)
        if arr.size:
            _ck(h.Dwrite(d, h5type, 0, 0, 0, arr.ctypes.data_as(C.c_void_p)), "H5Dwrite " + name)
        self.attr_str(d, "MATLAB_class", matlab_class)
        for k, v, t in extra:
            self.attr_num(d, k, v, t)
        h.Dclose(d); h.Sclose(sp)
        if dcpl:
            h.Pclose(dcpl)

    def numeric(self, parent, name, a):
        a = np.asarray(a)
        logical = a.dtype == np.bool_
        if logical:
            a = a.astype(np.uint8)
        elif a.dtype.kind == "c":
            raise Mat73Error("complex arrays are not part of the result dictionaries (%s)" % name)
        elif a.dtype not in _MATLAB_CLASS:
            a = a.astype(np.float64)
        if a.ndim == 0:
            a = a.reshape(1, 1)
        elif a.ndim == 1:
            a = a.reshape(1, -1)   # oned_as='row'
        cls = "logical" if logical else _MATLAB_CLASS[a.dtype]
        extra = [("MATLAB_int_decode", 1, np.int32)] if logical else []
        if a.size == 0:   # the shape instead of the data
            shape = np.array(a.shape, dtype=np.uint64)
            self.dataset(parent, name, shape, self.h.T[np.dtype("uint64")], cls, extra + [("MATLAB_empty", 1, np.uint8)])
            return
        self.dataset(parent, name, np.ascontiguousarray(a.T), self.h.T[a.dtype], cls, extra)

    def string(self, parent, name, text):
        units = np.frombuffer(text.encode("utf-16-le"), dtype=np.uint16)
        if units.size == 0:
            self.dataset(parent, name, np.array([0, 0], dtype=np.uint64), self.h.T[np.dtype("uint64")], "char", [("MATLAB_empty", 1, np.uint8)])
            return
        self.dataset(parent, name, np.ascontiguousarray(units.reshape(-1, 1)), self.h.T[np.dtype("uint16")], "char", [("MATLAB_int_decode", 2, np.int32)])

    def refs_group(self):
        if self.refs is None:
            h = self.h
            self.refs = _ck(h.Gcreate(self.fid, b"#refs#", 0, 0, 0), "H5Gcreate #refs#")
            # the canonical empty every cell-carrying v7.3 file has as "#refs#/a"
            self.dataset(self.refs, "a", np.array([0, 0], dtype=np.uint64), h.T[np.dtype("uint64")], "canonical empty", [("MATLAB_empty", 1, np.uint8)])
        return self.refs

    def ref_name(self):
        self.nref += 1
        n, out = self.nref, ""
        while True:   # b, c, ..., z, ba, bb, ...
            out = chr(ord("a") + n % 26) + out
            n //= 26
            if n == 0:
                return out if out != "a" else "aa"

    def cell(self, parent, name, items, shape=None):
        h = self.h
        refs = self.refs_group()
        n = len(items)
        out = np.zeros(n, dtype=np.uint64)
        for k, it in enumerate(items):
            rn = self.ref_name()
            self.write(refs, rn, it)
            _ck(h.Rcreate(out[k:].ctypes.data_as(C.c_void_p), refs, rn.encode(), H5R_OBJECT, -1), "H5Rcreate")
        if n == 0:
            self.dataset(parent, name, np.array([0, 0], dtype=np.uint64), h.T[np.dtype("uint64")], "cell", [("MATLAB_empty", 1, np.uint8)])
            return
        mshape = (1, n) if shape is None or len(shape) < 2 else tuple(shape)
        # references in column-major order of the MATLAB cell: file dims reversed
        filed = np.ascontiguousarray(out.reshape(mshape).T)
        self.dataset(parent, name, filed, h.T_REF, "cell")

    def struct(self, parent, name, d):
        h = self.h
        g = _ck(h.Gcreate(parent, name.encode("utf-8"), 0, 0, 0), "H5Gcreate " + name)
        self.attr_str(g, "MATLAB_class", "struct")
        keys = [str(k) for k in d]
        if keys and all(_FIELD_RE.match(k) for k in keys):
            self.attr_fields(g, keys)
        for k, v in d.items():
            self.write(g, str(k), v)
        h.Gclose(g)

    def write(self, parent, name, v):
        if "/" in name:
            raise Mat73Error("'/' in a field name (%r): not representable as an HDF5 link name" % name)
        if isinstance(v, dict):
            self.struct(parent, name, v)
        elif isinstance(v, str):
            self.string(parent, name, v)
        elif isinstance(v, bytes):
            self.string(parent, name, v.decode("utf-8"))
        elif isinstance(v, (list, tuple)):
            self.cell(parent, name, list(v))
        elif isinstance(v, np.ndarray) and v.dtype == object:
            self.cell(parent, name, list(v.reshape(-1)), v.shape)
        elif v is None:
            self.numeric(parent, name, np.zeros((0, 0)))
        else:
            self.numeric(parent, name, v)


def _header(platform=None, now=None):
    now = now or datetime.datetime.now()
    platform = platform or "CPython %d.%d.%d" % sys.version_info[:3]
    s = "MATLAB 7.3 MAT-file, Platform: %s, Created on: %s HDF5 schema 1.00 ." % (platform, now.strftime("%a %b %d %H:%M:%S %Y"))
    s = s[:116]
    return s.encode("ascii") + b" " * (116 - len(s)) + bytes(8) + bytes([0x00, 0x02]) + b"IM"


def savemat(file_name: str, mdict: dict, compress: bool = True, appendmat: bool = True) -> str:
    """Write `mdict` (str -> dict / list / ndarray / scalar / str, nested) as a MAT v7.3 file; returns the file name used.
    An existing file is replaced (the reference passes truncate_existing=True)."""
    h = _need()
    if appendmat and not file_name.endswith(".mat"):
        file_name += ".mat"
    for k in mdict:
        if not isinstance(k, str):
            raise Mat73Error("variable names are strings")
    fcpl = _ck(h.Pcreate(h.P_FILE_CREATE), "H5Pcreate")
    _ck(h.Pset_userblock(fcpl, 512), "H5Pset_userblock")
    fid = h.Fcreate(os.fsencode(file_name), H5F_ACC_TRUNC, fcpl, 0)
    h.Pclose(fcpl)
    _ck(fid, "H5Fcreate " + file_name)
    w = _Writer(h, fid, compress)
    try:
        for k, v in mdict.items():
            w.write(fid, k, v)
    finally:
        if w.refs is not None:
            h.Gclose(w.refs)
        h.Fclose(fid)
    with open(file_name, "r+b") as f:
        f.write(_header())
    return file_name


# ---------------------------------------------------------------------------------------------------------------------
class _Reader:
    def __init__(self, h, fid, squeeze):
        self.h, self.fid, self.squeeze = h, fid, squeeze

    def attr(self, obj, name):
        """a string or number attribute, or None"""
        h = self.h
        if h.Aexists(obj, name.encode()) <= 0:
            return None
        a = _ck(h.Aopen(obj, name.encode(), 0), "H5Aopen " + name)
        t = h.Aget_type(a)
        try:
            cls = h.Tget_class(t)
            if cls == H5T_STRING:
                if h.Tis_vstr(t) > 0:
                    ptr = C.c_char_p()
                    _ck(h.Aread(a, t, C.byref(ptr)), "H5Aread " + name)
                    return (ptr.value or b"").decode("utf-8", "replace")   # (a few bytes owned by the library are left to it)
                n = h.Tget_size(t)
                buf = C.create_string_buffer(n + 1)
                _ck(h.Aread(a, t, buf), "H5Aread " + name)
                return buf.raw[:n].split(b"\0")[0].decode("utf-8", "replace")
            if cls in (H5T_INTEGER, H5T_FLOAT, H5T_ENUM):
                v = np.zeros(1, dtype=np.int64 if cls != H5T_FLOAT else np.float64)
                _ck(h.Aread(a, h.T[v.dtype], v.ctypes.data_as(C.c_void_p)), "H5Aread " + name)
                return v[0].item()
            return None
        finally:
            h.Tclose(t); h.Aclose(a)

    def members(self, g):
        h = self.h
        info = _GInfo()
        _ck(h.Gget_info(g, C.byref(info)), "H5Gget_info")
        names = []
        for k in range(info.nlinks):
            n = _ck(h.Lget_name_by_idx(g, b".", H5_INDEX_NAME, H5_ITER_INC, k, None, 0, 0), "H5Lget_name_by_idx")
            buf = C.create_string_buffer(n + 1)
            h.Lget_name_by_idx(g, b".", H5_INDEX_NAME, H5_ITER_INC, k, buf, n + 1, 0)
            names.append(buf.value.decode("utf-8"))
        return names

    def group(self, g, top=False):
        names = self.members(g)
        order = None
        if self.h.Aexists(g, b"MATLAB_fields") > 0:
            order = self.fields(g)
        if order and set(order) == set(names):
            names = order
        out = {}
        for n in names:
            if top and n == "#refs#":
                continue
            o = _ck(self.h.Oopen(g, n.encode("utf-8"), 0), "H5Oopen " + n)
            try:
                out[n] = self.obj(o)
            finally:
                self.h.Oclose(o)
        return out

    def fields(self, g):
        h = self.h
        a = h.Aopen(g, b"MATLAB_fields", 0)
        if a < 0:
            return None
        t, sp = h.Aget_type(a), h.Aget_space(a)
        try:
            n = h.Sget_npoints(sp)
            arr = (_Hvl * n)()
            if h.Aread(a, t, arr) < 0:
                return None
            out = [C.string_at(arr[k].p, arr[k].len).decode("ascii", "replace") for k in range(n)]
            h.Dvlen_reclaim(t, sp, 0, arr)
            return out
        finally:
            h.Sclose(sp); h.Tclose(t); h.Aclose(a)

    def obj(self, o):
        kind = self.h.Iget_type(o)
        if kind == H5I_GROUP:
            return self.group(o)
        if kind == H5I_DATASET:
            return self.dataset(o)
        raise Mat73Error("unsupported HDF5 object type %d" % kind)

    def dataset(self, d):
        h = self.h
        cls = self.attr(d, "MATLAB_class") or ""
        sp, t = h.Dget_space(d), h.Dget_type(d)
        try:
            nd = h.Sget_ndims(sp)
            dims = (_hsize * max(nd, 1))()
            if nd > 0:
                h.Sget_dims(sp, dims, None)
            shape = tuple(int(x) for x in dims[:nd])
            tcls = h.Tget_class(t)
            if tcls == H5T_REFERENCE:
                refs = np.zeros(shape, dtype=np.uint64)
                _ck(h.Dread(d, h.T_REF, 0, 0, 0, refs.ctypes.data_as(C.c_void_p)), "H5Dread")
                flat = refs.T.reshape(-1)   # MATLAB order, then row-major over MATLAB's dims (a 1 x n cell: its n elements in order)
                items = []
                for r in flat:
                    rr = np.array([r], dtype=np.uint64)
                    o = _ck(h.Rdereference(d, 0, H5R_OBJECT, rr.ctypes.data_as(C.c_void_p)), "H5Rdereference")
                    try:
                        items.append(self.obj(o))
                    finally:
                        h.Oclose(o)
                return items
            if tcls not in (H5T_INTEGER, H5T_FLOAT, H5T_ENUM):
                raise Mat73Error("unsupported dataset type class %d" % tcls)
            nt = h.Tget_native(t, H5T_DIR_ASCEND)
            size = h.Tget_size(nt)
            if tcls == H5T_FLOAT:
                dt = np.dtype("float%d" % (8 * size))
            else:
                dt = np.dtype(("int%d" if h.Tget_sign(nt) else "uint%d") % (8 * size))
            buf = np.zeros(shape, dtype=dt)
            if buf.size:
                _ck(h.Dread(d, nt, 0, 0, 0, buf.ctypes.data_as(C.c_void_p)), "H5Dread")
            h.Tclose(nt)
            if self.attr(d, "MATLAB_empty"):
                mshape = tuple(int(x) for x in buf.reshape(-1))
                if cls == "char":
                    return ""
                if cls == "cell":
                    return []
                if self.squeeze:
                    mshape = tuple(x for x in mshape if x != 1) or (0,)
                return np.zeros(mshape, dtype=np.dtype(cls) if cls in _MATLAB_CLASS.values() else np.float64)
            a = np.ascontiguousarray(buf.T)   # MATLAB's dimensions
            if cls == "char":
                rows = [r.astype(np.uint16).tobytes().decode("utf-16-le") for r in a.reshape(-1, a.shape[-1] if a.ndim else 1)]
                return rows[0] if len(rows) == 1 else rows
            if cls == "logical":
                a = a.astype(bool)
            return np.squeeze(a) if self.squeeze else a
        finally:
            h.Tclose(t); h.Sclose(sp)


def loadmat(file_name: str, squeeze: bool = True) -> dict:
    """MAT v7.3 file -> {variable: dict (struct) / list (cell) / str (char) / ndarray}."""
    h = _need()
    if not is_v73(file_name):
        raise Mat73Error("%s is not a MAT v7.3 file" % file_name)
    fid = _ck(h.Fopen(os.fsencode(file_name), H5F_ACC_RDONLY, 0), "H5Fopen " + file_name)
    try:
        return _Reader(h, fid, squeeze).group(fid, top=True)
    finally:
        h.Fclose(fid)


def is_v73(file_name: str) -> bool:
    """MAT header of a v7.3 file, with the HDF5 signature behind the 512-byte user block"""
    with open(file_name, "rb") as f:
        head = f.read(520)
    return len(head) >= 520 and head.startswith(b"MATLAB 7.3 MAT-file") and head[512:520] == b"\x89HDF\r\n\x1a\n"
