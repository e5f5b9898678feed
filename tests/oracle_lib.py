"""ctypes access to the CPU oracle (oracle/kinodyn_oracle.cpp).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

from hippopt_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "oracle", "_build", "libkinodyn_oracle.so")


def build(force=False):
    src = [os.path.join(ROOT, "oracle", f) for f in ("kinodyn_oracle.cpp", "pose_oracle.cpp", "kinodyn_formulas.hpp", "scalar_types.hpp")]
    if force or not os.path.exists(SO) or any(os.path.getmtime(s) > os.path.getmtime(SO) for s in src):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    return SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.oracle_create.restype = C.c_void_p
        _lib.oracle_create.argtypes = [C.POINTER(_abi.DescC)]
        _lib.oracle_destroy.argtypes = [C.c_void_p]
        _lib.oracle_cost_term_name.restype = C.c_char_p
        _lib.oracle_pose_create.restype = C.c_void_p
        _lib.oracle_pose_create.argtypes = [C.POINTER(_abi.PoseDescC)]
        _lib.oracle_pose_destroy.argtypes = [C.c_void_p]
        _lib.oracle_pose_cost_term_name.restype = C.c_char_p
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


class Oracle:
    def __init__(self, settings, model):
        self.desc = _abi.DescC()
        self.desc.settings = settings.to_c()
        self.desc.model = model.to_c()
        self.desc.batch = 1
        self.desc.knot_begin = 0
        self.desc.knot_end = settings.horizon_length
        self.h = lib().oracle_create(C.byref(self.desc))
        assert self.h
        n, m, nnz, npar = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        lib().oracle_dims(C.c_void_p(self.h), C.byref(n), C.byref(m), C.byref(nnz), C.byref(npar))
        self.n, self.m, self.nnz, self.np = n.value, m.value, nnz.value, npar.value

    def __del__(self):
        if getattr(self, "h", None):
            lib().oracle_destroy(C.c_void_p(self.h))
            self.h = None

    def sparsity(self):
        ir = np.zeros(self.nnz, np.int32)
        jc = np.zeros(self.nnz, np.int32)
        lib().oracle_sparsity(C.c_void_p(self.h), ir.ctypes.data_as(C.POINTER(C.c_int)), jc.ctypes.data_as(C.POINTER(C.c_int)))
        return ir, jc

    def bounds(self, p):
        lb, ub = np.zeros(self.m), np.zeros(self.m)
        lib().oracle_bounds(C.c_void_p(self.h), _dp(np.ascontiguousarray(p)), _dp(lb), _dp(ub))
        return lb, ub

    def eval_fg(self, x, p):
        f = C.c_double()
        g = np.zeros(self.m)
        lib().oracle_eval_fg(C.c_void_p(self.h), _dp(np.ascontiguousarray(x)), _dp(np.ascontiguousarray(p)), C.byref(f), _dp(g))
        return f.value, g

    def eval(self, x, p):
        f = C.c_double()
        grad, g, jac = np.zeros(self.n), np.zeros(self.m), np.zeros(self.nnz)
        lib().oracle_eval(C.c_void_p(self.h), _dp(np.ascontiguousarray(x)), _dp(np.ascontiguousarray(p)),
                          C.byref(f), _dp(grad), _dp(g), _dp(jac))
        return f.value, grad, g, jac

    def cost_terms(self):
        out = np.zeros(_abi.NCOST_TERMS)
        lib().oracle_cost_terms(C.c_void_p(self.h), _dp(out))
        return out

    def hess(self, x, p, sigma, lam):
        """Exact Hessian of sigma f + lam^T g: numerically nonzero LOWER-triangle entries (rows, cols, vals), sorted by (col, row)."""
        fn = lib().oracle_eval_hess
        fn.restype = C.c_long
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        x = np.ascontiguousarray(x, np.float64); p = np.ascontiguousarray(p, np.float64); lam = np.ascontiguousarray(lam, np.float64)
        assert lam.size == self.m
        cnt = fn(self.h, x.ctypes.data, p.ctypes.data, float(sigma), lam.ctypes.data, None, None, None)
        rows, cols, vals = np.zeros(cnt, np.int32), np.zeros(cnt, np.int32), np.zeros(cnt)
        fn(self.h, x.ctypes.data, p.ctypes.data, float(sigma), lam.ctypes.data, rows.ctypes.data, cols.ctypes.data, vals.ctypes.data)
        return rows, cols, vals

    def row_blocks(self):
        out = []
        nb = lib().oracle_num_row_blocks(C.c_void_p(self.h))
        for i in range(nb):
            name = C.c_char_p()
            a, b, c, d = C.c_int(), C.c_int(), C.c_int(), C.c_int()
            lib().oracle_row_block(C.c_void_p(self.h), i, C.byref(name), C.byref(a), C.byref(b), C.byref(c), C.byref(d))
            out.append((name.value.decode(), a.value, b.value, c.value, d.value))
        return out


class PoseOracle:
    """oracle/pose_oracle.cpp: the static pose finder NLP (one pose)."""

    def __init__(self, settings, model):
        self.desc = _abi.PoseDescC()
        self.desc.settings = settings.to_c()
        self.desc.model = model.to_c()
        self.desc.batch = 1
        self.h = lib().oracle_pose_create(C.byref(self.desc))
        assert self.h
        n, m, nnz, npar = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        lib().oracle_pose_dims(C.c_void_p(self.h), C.byref(n), C.byref(m), C.byref(nnz), C.byref(npar))
        self.n, self.m, self.nnz, self.np = n.value, m.value, nnz.value, npar.value

    def __del__(self):
        if getattr(self, "h", None):
            lib().oracle_pose_destroy(C.c_void_p(self.h))
            self.h = None

    def sparsity(self):
        ir, jc = np.zeros(self.nnz, np.int32), np.zeros(self.nnz, np.int32)
        lib().oracle_pose_sparsity(C.c_void_p(self.h), ir.ctypes.data_as(C.POINTER(C.c_int)), jc.ctypes.data_as(C.POINTER(C.c_int)))
        return ir, jc

    def bounds(self, p):
        lb, ub = np.zeros(self.m), np.zeros(self.m)
        lib().oracle_pose_bounds(C.c_void_p(self.h), _dp(np.ascontiguousarray(p)), _dp(lb), _dp(ub))
        return lb, ub

    def eval(self, x, p):
        f = C.c_double()
        grad, g, jac = np.zeros(self.n), np.zeros(self.m), np.zeros(self.nnz)
        lib().oracle_pose_eval(C.c_void_p(self.h), _dp(np.ascontiguousarray(x)), _dp(np.ascontiguousarray(p)),
                               C.byref(f), _dp(grad), _dp(g), _dp(jac))
        return f.value, grad, g, jac

    def hess(self, x, p, sigma, lam):
        """dense [81][81] Hessian of sigma f + lam^T g (forward-over-forward AD)"""
        H = np.zeros((self.n, self.n))
        lib().oracle_pose_hess(C.c_void_p(self.h), _dp(np.ascontiguousarray(x)), _dp(np.ascontiguousarray(p)), C.c_double(sigma),
                               _dp(np.ascontiguousarray(lam, dtype=float)), _dp(H))
        return H

    def cost_terms(self):
        out = np.zeros(_abi.POSE_NCOST_TERMS)
        lib().oracle_pose_cost_terms(C.c_void_p(self.h), _dp(out))
        return out

    def cost_term_names(self):
        return [lib().oracle_pose_cost_term_name(i).decode() for i in range(_abi.POSE_NCOST_TERMS)]

    def row_blocks(self):
        out = []
        for i in range(lib().oracle_pose_num_row_blocks(C.c_void_p(self.h))):
            name = C.c_char_p()
            a, b = C.c_int(), C.c_int()
            lib().oracle_pose_row_block(C.c_void_p(self.h), i, C.byref(name), C.byref(a), C.byref(b))
            out.append((name.value.decode(), a.value, b.value))
        return out
