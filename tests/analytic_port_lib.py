"""ctypes access to the CPU baseline oracle/analytic_port.cpp (the engine's analytic knot program on the host, OpenMP over
knots).  TEST INFRASTRUCTURE / bench.py's cpu_baseline leg only."""
import ctypes as C
import os
import subprocess

import numpy as np

from hippopt_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "oracle", "_build", "libanalytic_port.so")


def build(native=False):
    """native: rebuild with -march=native for THIS machine's cores (bench.py on the GPU box) into a separate file."""
    if not native:
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "port"], stderr=subprocess.DEVNULL)
        return SO
    out = os.path.join(ROOT, "oracle", "_build", "libanalytic_port_native.so")
    subprocess.check_call(["make", "-s", "-B", "-C", os.path.join(ROOT, "oracle"), "port", "PORT_MARCH=native", "PORT_OUT=_build/libanalytic_port_native.so"],
                          stderr=subprocess.DEVNULL)
    return out


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class AnalyticPort:
    def __init__(self, settings, model, native=False):
        self.lib = C.CDLL(build(native))
        self.lib.port_create.restype = C.c_void_p
        self.desc = _abi.DescC()
        self.desc.settings = settings.to_c()
        self.desc.model = model.to_c()
        self.desc.batch = 1
        self.desc.knot_begin, self.desc.knot_end = 0, settings.horizon_length
        err = C.create_string_buffer(256)
        self.h = self.lib.port_create(C.byref(self.desc), err, 256)
        if not self.h:
            raise RuntimeError(err.value.decode())
        n, m, nnz = C.c_int(), C.c_int(), C.c_int()
        self.lib.port_dims(C.c_void_p(self.h), C.byref(n), C.byref(m), C.byref(nnz))
        self.n, self.m, self.nnz = n.value, m.value, nnz.value
        self.max_threads = int(self.lib.port_max_threads())
        self._out = (np.zeros(self.n), np.zeros(self.m), np.zeros(self.nnz))

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.port_destroy(C.c_void_p(self.h))
            self.h = None

    def set_params(self, p):
        self.lib.port_set_params(C.c_void_p(self.h), _dp(np.ascontiguousarray(p, dtype=np.float64)))

    def eval(self, x, threads=1):
        """(f, grad, g, jac); the arrays are reused from call to call (as IPOPT's are)"""
        f = C.c_double()
        grad, g, jac = self._out
        rc = self.lib.port_eval(C.c_void_p(self.h), _dp(np.ascontiguousarray(x, dtype=np.float64)), C.byref(f), _dp(grad), _dp(g), _dp(jac), int(threads))
        if rc != 0:
            raise RuntimeError("port_eval failed (%d)" % rc)
        return f.value, grad, g, jac
