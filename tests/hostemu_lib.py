"""ctypes access to the TEST-ONLY host emulation of the kernel's knot program (tests/hostemu)."""
import ctypes as C
import os
import subprocess

import numpy as np

from hippopt_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tests", "_build", "libhipnlp_hostemu.so")
SRC = [os.path.join(ROOT, "tests", "hostemu", "hostemu.cpp")] + [
    os.path.join(ROOT, "hippopt_amd", "csrc", f) for f in ("layout.h", "knot_body.h", "nlp_defs.h", "pose_body.h", "pose_hess_body.h", "pose_layout.h", "knot_hess_body.h", "knot_hess_layout.h", "knot_hess_terrain.h", "knot_tanh.h")]


def build():
    if not os.path.exists(SO) or any(os.path.getmtime(s) > os.path.getmtime(SO) for s in SRC):
        os.makedirs(os.path.dirname(SO), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra", "-o", SO, SRC[0]])
    return SO


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


class HostEmu:
    def __init__(self, settings, model, detect_simple_bounds=False, jac_varying_first=False):
        self.lib = C.CDLL(build())
        self.lib.hostemu_create.restype = C.c_void_p
        self.desc = _abi.DescC()
        self.desc.settings = settings.to_c()
        self.desc.model = model.to_c()
        self.desc.batch = 1
        self.desc.knot_begin, self.desc.knot_end = 0, settings.horizon_length
        self.desc.flags = (_abi.FLAG_DETECT_SIMPLE_BOUNDS if detect_simple_bounds else 0) | (_abi.FLAG_JAC_VARYING_FIRST if jac_varying_first else 0)
        err = C.create_string_buffer(256)
        self.h = self.lib.hostemu_create(C.byref(self.desc), err, 256)
        if not self.h:
            raise RuntimeError(err.value.decode())
        n, m, nnz = C.c_int(), C.c_int(), C.c_int()
        self.lib.hostemu_dims(C.c_void_p(self.h), C.byref(n), C.byref(m), C.byref(nnz))
        self.n, self.m, self.nnz = n.value, m.value, nnz.value

    def sparsity(self):
        ir, jc = np.zeros(self.nnz, np.int32), np.zeros(self.nnz, np.int32)
        self.lib.hostemu_sparsity(C.c_void_p(self.h), _ip(ir), _ip(jc))
        return ir, jc

    def constant_mask(self):
        """bool [nnz] in pattern order: entries of jac g that do not depend on x (emit_jc in knot_body.h)"""
        mask = np.zeros(self.nnz, np.uint8)
        self.lib.hostemu_constant_mask(C.c_void_p(self.h), mask.ctypes.data_as(C.POINTER(C.c_ubyte)))
        return mask.astype(bool)

    def constant_fill(self, p, poison=np.nan):
        """[nnz]: the constant entries under the parameters p at their positions (what the library fills a host destination with),
        `poison` everywhere else"""
        out = np.zeros(self.nnz)
        self.lib.hostemu_constant_fill(C.c_void_p(self.h), _dp(np.ascontiguousarray(p, dtype=np.float64)), C.c_double(poison), _dp(out))
        return out

    def vary_counts(self):
        """(varying entries, all entries) of a first / interior / last knot block, constant entries of the whole pattern"""
        a, b, c = (C.c_int * 3)(), (C.c_int * 3)(), C.c_int()
        self.lib.hostemu_vary_counts(C.c_void_p(self.h), a, b, C.byref(c))
        return list(a), list(b), c.value

    def vary_partition(self):
        """(ok, smallest recorded slot that depends on x, largest, window begin, window end): Layout::vary_partition_ok and its evidence"""
        out = (C.c_int * 5)()
        self.lib.hostemu_vary_partition(C.c_void_p(self.h), out)
        return bool(out[0]), out[1], out[2], out[3], out[4]

    def hess_phases(self):
        """(phase of the Hessian program in which the entry at every position of a knot block is emitted, length of the early run, barrier
        behind which that run is final): HessLayout::pos_phase / early_run / early_phase"""
        n = self.lib.hostemu_hess_nnz(C.c_void_p(self.h))
        ir, jc = np.zeros(n, np.int32), np.zeros(n, np.int32)
        self.lib.hostemu_hess_sparsity(C.c_void_p(self.h), _ip(ir), _ip(jc))
        nnz_knot = int(((jc // 189) == 1).sum())
        ph, ep = np.zeros(nnz_knot, np.uint8), C.c_int()
        self.lib.hostemu_hess_phases.restype = C.c_int
        run = self.lib.hostemu_hess_phases(C.c_void_p(self.h), ph.ctypes.data_as(C.POINTER(C.c_ubyte)), C.byref(ep))
        return ph, int(run), int(ep.value), (ir[(jc // 189) == 1] % 189, jc[(jc // 189) == 1] % 189)

    def early_violations(self):
        """entries marked final after the second phase whose staged value changed later, in the last wave-order evaluation"""
        self.lib.hostemu_early_violations.restype = C.c_long
        return int(self.lib.hostemu_early_violations())

    def output_phases(self):
        """phase of the knot program (barriers passed) in which every entry of jac g (pattern order) / every row of g gets its value"""
        jp, gp = np.zeros(self.nnz, np.uint8), np.zeros(self.m, np.uint8)
        self.lib.hostemu_output_phases(C.c_void_p(self.h), jp.ctypes.data_as(C.c_void_p), gp.ctypes.data_as(C.c_void_p))
        return jp, gp

    def bounds(self, p):
        lb, ub = np.zeros(self.m), np.zeros(self.m)
        self.lib.hostemu_bounds(C.c_void_p(self.h), _dp(np.ascontiguousarray(p)), _dp(lb), _dp(ub))
        return lb, ub

    def bounds_x(self, p):
        lb, ub = np.zeros(self.n), np.zeros(self.n)
        self.lib.hostemu_bounds_x(C.c_void_p(self.h), _dp(np.ascontiguousarray(p)), _dp(lb), _dp(ub))
        return lb, ub

    def kept_rows(self):
        """row of this layout behind every row of the full subject_to list (-1: lifted into a bound)"""
        out = np.zeros(self.lib.hostemu_m_full(C.c_void_p(self.h)), np.int32)
        self.lib.hostemu_lift_map(C.c_void_p(self.h), _ip(out))
        return out

    def row_blocks(self):
        out = []
        for i in range(self.lib.hostemu_num_row_blocks(C.c_void_p(self.h))):
            name = C.c_char_p()
            a, b, c, d = C.c_int(), C.c_int(), C.c_int(), C.c_int()
            self.lib.hostemu_row_block(C.c_void_p(self.h), i, C.byref(name), C.byref(a), C.byref(b), C.byref(c), C.byref(d))
            out.append((name.value.decode(), a.value, b.value, c.value, d.value))
        return out

    def eval(self, x, p, compact=False):
        """compact: on the COMPACT scratch layout of the planar device kernel (arrays with disjoint lifetimes share storage)"""
        f = C.c_double()
        grad, g, jac, ct = np.zeros(self.n), np.full(self.m, np.nan), np.full(self.nnz, np.nan), np.zeros(_abi.NCOST_TERMS)
        fn = self.lib.hostemu_eval_compact if compact else self.lib.hostemu_eval
        rc = fn(C.c_void_p(self.h), _dp(np.ascontiguousarray(x)), _dp(np.ascontiguousarray(p)), C.byref(f), _dp(grad), _dp(g), _dp(jac), _dp(ct))
        if compact and rc != 0:
            raise RuntimeError("hostemu_eval_compact failed")
        return f.value, grad, g, jac, ct


def _eval_wave_order(self, x, p, waves, order):
    """the knot program with every phase's groups run wave by wave in a permuted wave order (hostemu_eval_wave_order)"""
    f = C.c_double()
    grad, g, jac, ct = np.zeros(self.n), np.full(self.m, np.nan), np.full(self.nnz, np.nan), np.zeros(_abi.NCOST_TERMS)
    rc = self.lib.hostemu_eval_wave_order(C.c_void_p(self.h), _dp(np.ascontiguousarray(x)), _dp(np.ascontiguousarray(p)), int(waves), int(order),
                                          C.byref(f), _dp(grad), _dp(g), _dp(jac), _dp(ct))
    if rc != 0:
        raise RuntimeError("hostemu_eval_wave_order failed")
    return f.value, grad, g, jac, ct


HostEmu.eval_wave_order = _eval_wave_order


def set_wave_order(order):
    """order >= 0: the pose program and the two Hessian programs run every phase wave by wave in that order of the four waves
    (0 ascending, 1 descending, 2 rotated by two); -1: plain table order"""
    C.CDLL(build()).hostemu_set_wave_order(int(order))   # (the same library object every CDLL of this path shares)


def _hostemu_hess_methods():
    def hess_sparsity(self):
        self.lib.hostemu_hess_nnz.restype = C.c_long
        nnz = self.lib.hostemu_hess_nnz(C.c_void_p(self.h))
        if nnz < 0:
            self.lib.hostemu_hess_error.restype = C.c_char_p
            raise RuntimeError(self.lib.hostemu_hess_error(C.c_void_p(self.h)).decode())
        ir, jc = np.zeros(nnz, np.int32), np.zeros(nnz, np.int32)
        self.lib.hostemu_hess_sparsity(C.c_void_p(self.h), _ip(ir), _ip(jc))
        return ir, jc

    def hess(self, x, p, sigma, lam):
        ir, _ = self.hess_sparsity()
        out = np.full(ir.size, np.nan)
        self.lib.hostemu_hess(C.c_void_p(self.h), _dp(np.ascontiguousarray(x)), _dp(np.ascontiguousarray(p)), C.c_double(sigma),
                              _dp(np.ascontiguousarray(lam, np.float64)), _dp(out))
        return out
    HostEmu.hess_sparsity = hess_sparsity
    HostEmu.hess = hess


_hostemu_hess_methods()


class PoseHostEmu:
    """Host emulation of the pose program (pose_body.h) + the pose layout tables."""

    def __init__(self, settings, model, detect_simple_bounds=False):
        self.lib = C.CDLL(build())
        self.lib.hostemu_pose_create.restype = C.c_void_p
        self.desc = _abi.PoseDescC()
        self.desc.settings = settings.to_c()
        self.desc.model = model.to_c()
        self.desc.batch = 1
        err = C.create_string_buffer(256)
        self.h = self.lib.hostemu_pose_create(C.byref(self.desc), err, 256)
        if not self.h:
            raise RuntimeError(err.value.decode())
        n, m, nnz = C.c_int(), C.c_int(), C.c_int()
        self.lib.hostemu_pose_dims(C.c_void_p(self.h), C.byref(n), C.byref(m), C.byref(nnz))
        self.n, self.m, self.nnz = n.value, m.value, nnz.value

    def sparsity(self):
        ir, jc = np.zeros(self.nnz, np.int32), np.zeros(self.nnz, np.int32)
        self.lib.hostemu_pose_sparsity(C.c_void_p(self.h), _ip(ir), _ip(jc))
        return ir, jc

    def bounds(self, p):
        lb, ub = np.zeros(self.m), np.zeros(self.m)
        self.lib.hostemu_pose_bounds(C.c_void_p(self.h), _dp(np.ascontiguousarray(p)), _dp(lb), _dp(ub))
        return lb, ub

    def row_blocks(self):
        out = []
        for i in range(self.lib.hostemu_pose_num_row_blocks(C.c_void_p(self.h))):
            name = C.c_char_p()
            a, b = C.c_int(), C.c_int()
            self.lib.hostemu_pose_row_block(C.c_void_p(self.h), i, C.byref(name), C.byref(a), C.byref(b))
            out.append((name.value.decode(), a.value, b.value))
        return out

    def eval(self, x, p):
        f = C.c_double()
        grad, g, jac, ct = np.zeros(self.n), np.full(self.m, np.nan), np.full(self.nnz, np.nan), np.zeros(_abi.POSE_NCOST_TERMS)
        self.lib.hostemu_pose_eval(C.c_void_p(self.h), _dp(np.ascontiguousarray(x)), _dp(np.ascontiguousarray(p)),
                                   C.byref(f), _dp(grad), _dp(g), _dp(jac), _dp(ct))
        return f.value, grad, g, jac, ct

    def map_violations(self):
        """entries the pose program sent through the device emitter's staging map (pose_body.h pjs) in a way the device kernel could not serve:
        a slot of the constant region through J, a slot outside the kept pieces through JC / JD or in the pattern (count since the library
        was loaded)"""
        self.lib.hostemu_pose_map_violations.restype = C.c_long
        return int(self.lib.hostemu_pose_map_violations())

    def hess_sparsity(self):
        n = C.c_int()
        self.lib.hostemu_pose_hess_dims(C.c_void_p(self.h), C.byref(n))
        ir, jc = np.zeros(n.value, np.int32), np.zeros(n.value, np.int32)
        self.lib.hostemu_pose_hess_sparsity(C.c_void_p(self.h), _ip(ir), _ip(jc))
        return ir, jc

    def hess(self, x, p, sigma, lam):
        ir, _ = self.hess_sparsity()
        out = np.full(ir.size, np.nan)
        self.lib.hostemu_pose_hess(C.c_void_p(self.h), _dp(np.ascontiguousarray(x)), _dp(np.ascontiguousarray(p)), C.c_double(sigma),
                                   _dp(np.ascontiguousarray(lam, dtype=float)), _dp(out))
        return out
