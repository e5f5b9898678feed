"""The method set the NLP drivers of HipNlpSolver use on a HipNlp handle, backed by the TEST-ONLY host emulation of the knot program
(no GPU in the build container): for tests of the host logic around the engine."""
import ctypes as C

import numpy as np

from hippopt_amd import _abi
from hostemu_lib import HostEmu


class EmuEngine:
    def __init__(self, st, model, p=None, detect_simple_bounds=False):
        """detect_simple_bounds: the emulation runs on the LIFTED layout (layout.h, Layout::build(..., lift)), as a handle created
        with HIPNLP_FLAG_DETECT_SIMPLE_BOUNDS does"""
        self.he, self.p = HostEmu(st, model, detect_simple_bounds=detect_simple_bounds), p
        self._full = HostEmu(st, model) if detect_simple_bounds else self.he
        self.lifted = bool(detect_simple_bounds)
        self.n, self.m, self.nnz = self.he.n, self.he.m, self.he.nnz
        self.m_full = self._full.m
        self.evaluations = 0
        self.params_generation = 0
        self._terms = np.zeros((1, _abi.NCOST_TERMS))

    def set_params(self, p):
        self.p = np.asarray(p, float).reshape(-1)
        self.params_generation += 1

    def simple_rows(self):
        a, b = np.zeros(self.m_full, np.int32), np.zeros(self.m_full, np.int32)
        self.he.lib.hostemu_simple_rows(C.c_void_p(self.he.h), a.ctypes.data_as(C.POINTER(C.c_int)), b.ctypes.data_as(C.POINTER(C.c_int)))
        return a, b

    def lift_map(self, with_bounds=True):
        lb, ub = self._full.bounds(self.p) if with_bounds else (None, None)
        return self.he.kept_rows(), lb, ub

    def sparsity(self):
        return self.he.sparsity()

    def row_blocks(self):
        return self.he.row_blocks()

    def bounds(self):
        lb, ub = self.he.bounds(self.p)
        lbx, ubx = self.he.bounds_x(self.p)
        return lbx, ubx, lb, ub

    def eval(self, x, new_x=True, want=None, nan_ok=False):
        f, grad, g, jac, terms = self.he.eval(np.asarray(x).reshape(-1), self.p)
        self.evaluations += 1
        self._terms = terms[None, :]
        return np.array([f]), grad[None], g[None], jac[None]

    def cost_terms(self):
        names = ["term%d" % i for i in range(_abi.NCOST_TERMS)]
        return names, self._terms
