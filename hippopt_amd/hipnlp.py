"""ctypes binding of the C-ABI in include/hipnlp.h (libhipnlp.so, built by __graft_entry__.build()).

There is no CPU fallback: loading fails loudly when the library is missing, and `HipNlp(...)`
raises when no HIP device is present.
"""
import ctypes as C
import weakref
import os
import sys as _sys

import numpy as np

from . import _abi

# (HIPNLP_LIB_PATH: diagnostic override to load another BUILD of the same HIP library, e.g. an A/B variant under tools/diag/_build)
_LIB_PATH = os.environ.get("HIPNLP_LIB_PATH") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libhipnlp.so")
_libs = {}     # path -> bound library (the product library; tests also load the diagnostic build of the same sources beside it)

EXPORTS = [
    "hipnlp_abi_version", "hipnlp_build_info",
    "hipnlp_create", "hipnlp_destroy", "hipnlp_last_error", "hipnlp_get_dims", "hipnlp_set_params",
    "hipnlp_bounds", "hipnlp_simple_rows", "hipnlp_lift_map", "hipnlp_sparsity", "hipnlp_eval", "hipnlp_eval_device",
    "hipnlp_cost_terms", "hipnlp_cost_term_name", "hipnlp_num_row_blocks", "hipnlp_row_block",
    "hipnlp_last_kernel_ms", "hipnlp_profile_begin", "hipnlp_profile_end", "hipnlp_kernels_per_eval", "hipnlp_profile_begin_runs",
    "hipnlp_eval_device_shard", "hipnlp_stage_rows", "hipnlp_reassemble",
    "hipnlp_jac_vary_layout", "hipnlp_fill_jac_constants", "hipnlp_eval_device_vary", "hipnlp_eval_device_shard_vary", "hipnlp_eval_device_peers_vary", "hipnlp_reassemble_scatter",
    "hipnlp_ipc_alloc", "hipnlp_ipc_open", "hipnlp_ipc_close", "hipnlp_ipc_free", "hipnlp_peer_push", "hipnlp_peer_signal", "hipnlp_peer_signal_checked", "hipnlp_peer_wait", "hipnlp_eval_device_peers",
    "hipnlp_device_numa_node", "hipnlp_pin_thread_to_device_numa_node",
    "hipnlp_hess_nnz", "hipnlp_hess_sparsity", "hipnlp_eval_hess", "hipnlp_eval_hess_at", "hipnlp_set_hessian_early_run", "hipnlp_get_hessian_early_run", "hipnlp_hessian_early_run_reason", "hipnlp_eval_hess_device",
    "hipnlp_eval_pinned", "hipnlp_set_prefetch", "hipnlp_set_early_outputs", "hipnlp_set_host_timing", "hipnlp_host_register", "hipnlp_host_unregister",
    "hipnlp_multi_create", "hipnlp_multi_plan", "hipnlp_multi_info", "hipnlp_multi_breakdown", "hipnlp_multi_set_threads",
    "hipnlp_host_breakdown", "hipnlp_set_auto_register", "hipnlp_host_stats", "hipnlp_set_constant_jacobian", "hipnlp_forget_jac_destination", "hipnlp_jac_constant_mask", "hipnlp_host_release_auto_ranges",
    "hipnlp_pose_create", "hipnlp_pose_destroy", "hipnlp_pose_last_error", "hipnlp_pose_get_dims", "hipnlp_pose_set_params",
    "hipnlp_pose_bounds", "hipnlp_pose_sparsity", "hipnlp_pose_eval", "hipnlp_pose_eval_device", "hipnlp_pose_cost_terms",
    "hipnlp_pose_cost_term_name", "hipnlp_pose_num_row_blocks", "hipnlp_pose_row_block", "hipnlp_pose_last_kernel_ms", "hipnlp_pose_set_host_timing",
    "hipnlp_pose_hess_nnz", "hipnlp_pose_hess_sparsity", "hipnlp_pose_eval_hess", "hipnlp_pose_eval_hess_device",
    # include/hipnlp_ipopt.h: IPOPT's C callback quartet (+ eval_h) on top of the functions above
    "hipnlp_ipopt_eval_f", "hipnlp_ipopt_eval_grad_f", "hipnlp_ipopt_eval_g", "hipnlp_ipopt_eval_jac_g", "hipnlp_ipopt_eval_h",
    "hipnlp_ipopt_sizes", "hipnlp_ipopt_bounds", "hipnlp_ipopt_attach", "hipnlp_ipopt_detach", "hipnlp_ipopt_set_early_outputs",
]
G_STAGE = 550


class HipNlpError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"hipnlp error {code}: {message}")
        self.code = code


def library_path():
    return _LIB_PATH


def build_info():
    """how the loaded library was built (hipnlp_build_info): recorded by bench.py beside every measurement"""
    return load_library().hipnlp_build_info().decode()


def parse_cpulist(text):
    """'0-63,128-191' -> the set of CPU numbers (the format of /sys/devices/system/node/node<N>/cpulist)"""
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def device_numa_node(device=0):
    """NUMA node of the host that the card hangs off (hipnlp_device_numa_node), None when the system does not say"""
    node = C.c_int(-1)
    lib = load_library()
    lib.hipnlp_device_numa_node.argtypes = [C.c_int, C.POINTER(C.c_int)]
    if lib.hipnlp_device_numa_node(int(device), C.byref(node)) != 0 or node.value < 0:
        return None
    return node.value


def pin_to_device_numa_node(device=0):
    """Restrict the calling thread to the CPUs of the card's NUMA node (those of them it is allowed to run on).  The host-buffer paths are
    link-bound: the thread that calls them, and the arrays it allocates from then on (first touch), belong on the side of the host the
    card hangs off — 37.6 - 39.0 against 40.2 - 41.7 us per 100-knot hipnlp_eval with all four outputs, 55 against 60 - 62 us per exact
    Hessian on a two-socket host (profiles/r05_early_stores_by_box.txt).  Returns {"node", "cpus"} or None (node unknown / no CPU of it allowed:
    nothing changed)."""
    node = device_numa_node(device)
    if node is None:
        return None
    try:
        cpus = parse_cpulist(open("/sys/devices/system/node/node%d/cpulist" % node).read()) & set(os.sched_getaffinity(0))
    except (OSError, ValueError):
        return None
    if not cpus:
        return None
    os.sched_setaffinity(0, cpus)
    return {"node": node, "cpus": len(cpus)}


def multi_plan(horizon, n_shards, shard):
    """hipnlp_multi_plan (no device needed): ((knot_begin, knot_end), [(offset, count), ...] of x in doubles) of one shard of a multi-device handle"""
    kb, ke, rg = C.c_int32(), C.c_int32(), (C.c_int64 * 8)()
    rc = load_library().hipnlp_multi_plan(int(horizon), int(n_shards), int(shard), C.byref(kb), C.byref(ke), rg)
    if rc != 0:
        raise HipNlpError(rc, "hipnlp_multi_plan: bad arguments")
    return (kb.value, ke.value), [(int(rg[2 * i]), int(rg[2 * i + 1])) for i in range(4) if rg[2 * i + 1] > 0]


def load_library(path=None):
    """the bound C library.  path: another BUILD of the same sources (tests: the diagnostic build, tests/_build/libhipnlp_diag.so, whose
    hipnlp_create honours the HIPNLP_* environment overrides the shipped library does not read); default: the product library"""
    path = path or _LIB_PATH
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). The engine has no CPU fallback.")
    try:
        # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64; if libhipnlp.so pulled in /opt/rocm's copy first, a
        # later `import torch` would bind to that one and find no devices.  Loading torch first makes both share torch's.
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(path)
    dp, ip, vp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.c_void_p
    lib.hipnlp_create.argtypes = [C.POINTER(_abi.DescC), C.POINTER(vp)]
    lib.hipnlp_destroy.argtypes = [vp]
    lib.hipnlp_destroy.restype = None
    lib.hipnlp_last_error.argtypes = [vp]
    lib.hipnlp_last_error.restype = C.c_char_p
    lib.hipnlp_get_dims.argtypes = [vp, C.POINTER(_abi.DimsC)]
    lib.hipnlp_set_params.argtypes = [vp, dp]
    lib.hipnlp_bounds.argtypes = [vp, dp, dp, dp, dp]
    lib.hipnlp_simple_rows.argtypes = [vp, ip, ip]
    lib.hipnlp_lift_map.argtypes = [vp, ip, dp, dp]
    lib.hipnlp_build_info.restype = C.c_char_p
    if lib.hipnlp_abi_version() != _abi.ABI_VERSION:
        raise ImportError(f"{path} implements ABI version {lib.hipnlp_abi_version()}, this package expects {_abi.ABI_VERSION}: rebuild the library")
    lib.hipnlp_sparsity.argtypes = [vp, ip, ip]
    lib.hipnlp_eval.argtypes = [vp, dp, C.c_int, dp, dp, dp, dp]
    lib.hipnlp_eval_device.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    lib.hipnlp_cost_terms.argtypes = [vp, dp]
    lib.hipnlp_cost_term_name.argtypes = [C.c_int]
    lib.hipnlp_cost_term_name.restype = C.c_char_p
    lib.hipnlp_num_row_blocks.argtypes = [vp]
    lib.hipnlp_row_block.argtypes = [vp, C.c_int, C.POINTER(C.c_char_p), ip, ip, ip, ip]
    lib.hipnlp_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    lib.hipnlp_eval_device_shard.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    lib.hipnlp_stage_rows.argtypes = [vp, C.c_int, ip]
    lib.hipnlp_profile_begin.argtypes = [vp, C.c_int, C.c_int]
    lib.hipnlp_profile_end.argtypes = [vp, dp, dp, C.POINTER(C.c_int)]
    lib.hipnlp_pose_create.argtypes = [C.POINTER(_abi.PoseDescC), C.POINTER(vp)]
    lib.hipnlp_pose_destroy.argtypes = [vp]
    lib.hipnlp_pose_destroy.restype = None
    lib.hipnlp_pose_last_error.argtypes = [vp]
    lib.hipnlp_pose_last_error.restype = C.c_char_p
    lib.hipnlp_pose_get_dims.argtypes = [vp, C.POINTER(_abi.PoseDimsC)]
    lib.hipnlp_pose_set_params.argtypes = [vp, dp]
    lib.hipnlp_pose_bounds.argtypes = [vp, dp, dp]
    lib.hipnlp_pose_sparsity.argtypes = [vp, ip, ip]
    lib.hipnlp_pose_eval.argtypes = [vp, dp, dp, dp, dp, dp]
    lib.hipnlp_pose_eval_device.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    lib.hipnlp_pose_cost_terms.argtypes = [vp, dp]
    lib.hipnlp_pose_cost_term_name.argtypes = [C.c_int]
    lib.hipnlp_pose_cost_term_name.restype = C.c_char_p
    lib.hipnlp_pose_num_row_blocks.argtypes = [vp]
    lib.hipnlp_pose_row_block.argtypes = [vp, C.c_int, C.POINTER(C.c_char_p), ip, ip]
    lib.hipnlp_pose_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    lib.hipnlp_pose_hess_nnz.argtypes = [vp, ip]
    lib.hipnlp_pose_hess_sparsity.argtypes = [vp, ip, ip]
    lib.hipnlp_pose_eval_hess.argtypes = [vp, dp, dp, dp, dp]
    lib.hipnlp_pose_eval_hess_device.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.hipnlp_hess_nnz.argtypes = [vp, C.POINTER(C.c_int64)]
    lib.hipnlp_hess_sparsity.argtypes = [vp, ip, ip]
    lib.hipnlp_eval_hess.argtypes = [vp, dp, dp, dp, dp]
    lib.hipnlp_eval_hess_at.argtypes = [vp, dp, C.c_int, dp, dp, dp]
    lib.hipnlp_set_hessian_early_run.argtypes = [vp, C.c_int]
    lib.hipnlp_get_hessian_early_run.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), dp, dp]
    lib.hipnlp_hessian_early_run_reason.argtypes = [vp]
    lib.hipnlp_hessian_early_run_reason.restype = C.c_char_p
    lib.hipnlp_eval_hess_device.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.hipnlp_kernels_per_eval.argtypes = [vp]
    lib.hipnlp_profile_begin_runs.argtypes = [vp, C.c_int, C.c_int]
    lib.hipnlp_reassemble.argtypes = [vp, vp, vp, C.c_int64, C.c_int, C.c_int64, vp, vp]
    lib.hipnlp_eval_pinned.argtypes = [vp, dp, C.c_int, C.c_uint, C.POINTER(dp), C.POINTER(dp), C.POINTER(dp), C.POINTER(dp)]
    lib.hipnlp_set_prefetch.argtypes = [vp, C.c_uint]
    lib.hipnlp_set_early_outputs.argtypes = [vp, C.c_int]
    lib.hipnlp_set_host_timing.argtypes = [vp, C.c_int]
    lib.hipnlp_host_register.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    lib.hipnlp_host_unregister.argtypes = [vp]
    lib.hipnlp_host_breakdown.argtypes = [vp, dp]
    lib.hipnlp_set_auto_register.argtypes = [vp, C.c_int]
    lib.hipnlp_host_stats.argtypes = [vp, C.POINTER(C.c_long)]
    lib.hipnlp_set_constant_jacobian.argtypes = [vp, C.c_int]
    lib.hipnlp_jac_constant_mask.argtypes = [vp, C.POINTER(C.c_ubyte)]
    lib.hipnlp_forget_jac_destination.argtypes = [vp, vp]
    lib.hipnlp_jac_vary_layout.argtypes = [vp, C.POINTER(C.c_int64)]
    lib.hipnlp_fill_jac_constants.argtypes = [vp, vp, C.c_int, vp]
    lib.hipnlp_eval_device_vary.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    lib.hipnlp_eval_device_shard_vary.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    lib.hipnlp_eval_device_peers_vary.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp]
    lib.hipnlp_reassemble_scatter.argtypes = [vp, vp, vp, vp, C.c_int64, C.c_int, C.c_int64, vp, vp]
    lib.hipnlp_multi_create.argtypes = [C.POINTER(_abi.DescC), ip, C.c_int, C.POINTER(vp)]
    lib.hipnlp_multi_info.argtypes = [vp, ip, ip, ip, ip, ip]
    lib.hipnlp_multi_plan.argtypes = [C.c_int, C.c_int, C.c_int, ip, ip, C.POINTER(C.c_int64)]
    lib.hipnlp_multi_breakdown.argtypes = [vp, dp]
    lib.hipnlp_multi_set_threads.argtypes = [vp, C.c_int, C.c_double]
    _libs[path] = lib
    return lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


class HipNlp:
    """One engine handle: a kinodynamic NLP (settings + robot model) on one HIP device — or, `devices=[...]`, on several behind the
    same host-buffer calls (hipnlp_multi_create: one caller, the horizon cut into one contiguous knot range per entry)."""

    def __init__(self, settings, model, batch=1, knot_begin=0, knot_end=0, device=0, desc=None, detect_simple_bounds=False,
                 jac_varying_first=False, library=None, devices=None):
        """desc: a ready hipnlp_desc (e.g. hippopt_amd.from_reference.from_reference) instead of settings / model
        detect_simple_bounds: the handle is the REDUCED NLP nlpsol hands to IPOPT under Opti's {"detect_simple_bounds": True}
        (HIPNLP_FLAG_DETECT_SIMPLE_BOUNDS): single-variable rows are bounds on x, not rows of g
        jac_varying_first: HIPNLP_FLAG_JAC_VARYING_FIRST — inside a knot's block of jac g the entries that depend on x come first,
        the constant ones behind them (triplet consumers such as IPOPT; `sparsity()` returns that order)
        library: path of another build of the library (tests: the diagnostic build); default: the product library
        devices: HIP ordinals, one shard of the horizon per entry (an ordinal may repeat: its shards share the card) — the handle then
        serves eval / eval_pinned / eval_hess / cost_terms and the metadata calls; the device-pointer calls raise HipNlpError(-6)"""
        self.lib = load_library(library)
        if desc is not None:
            self.desc = desc
            batch = desc.batch
        else:
            self.desc = _abi.DescC()
            self.desc.settings = settings.to_c()
            self.desc.model = model.to_c()
            self.desc.batch = int(batch)
            self.desc.knot_begin, self.desc.knot_end = int(knot_begin), int(knot_end)
            self.desc.device = int(device)
            self.desc.flags = (_abi.FLAG_DETECT_SIMPLE_BOUNDS if detect_simple_bounds else 0) | (_abi.FLAG_JAC_VARYING_FIRST if jac_varying_first else 0)
        h = C.c_void_p()
        self.devices = None if devices is None else [int(v) for v in devices]
        if self.devices is not None:
            if knot_begin or knot_end:
                raise ValueError("devices=[...] cuts the whole horizon itself: no knot_begin / knot_end")
            dev = (C.c_int32 * len(self.devices))(*self.devices)
            rc = self.lib.hipnlp_multi_create(C.byref(self.desc), dev, len(self.devices), C.byref(h))
        else:
            rc = self.lib.hipnlp_create(C.byref(self.desc), C.byref(h))
        if rc != 0:
            raise HipNlpError(rc, self.lib.hipnlp_last_error(None).decode())
        self.h = h
        d = _abi.DimsC()
        self._check(self.lib.hipnlp_get_dims(self.h, C.byref(d)))
        self.dims = d
        self.batch = int(batch)
        self.n, self.m, self.nnz, self.np = d.n, d.m, d.nnz, d.np
        self.m_full, self.n_lifted = d.m_full, d.n_lifted
        self.lifted = bool(self.desc.flags & _abi.FLAG_DETECT_SIMPLE_BOUNDS)
        self.jac_varying_first = bool(self.desc.flags & _abi.FLAG_JAC_VARYING_FIRST)
        self.constants_in_place = self.jac_varying_first   # (hipnlp_create's default for hipnlp_set_constant_jacobian)
        self.params_generation = 0   # bumped by set_params (what caches of parameter-dependent data key on)
        self._addresses = {}
        self._h_value = h.value
        self._eval_raw = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)(("hipnlp_eval", self.lib))

    @classmethod
    def from_desc(cls, desc):
        """an engine from a hipnlp_desc structure or its raw bytes (a committed fixture, another process, ...)"""
        if not isinstance(desc, _abi.DescC):
            desc = _abi.DescC.from_buffer_copy(bytes(desc))
        return cls(None, None, desc=desc)

    def close(self):
        if getattr(self, "h", None):
            self.lib.hipnlp_destroy(self.h)
            self.h = None
            self._h_value = None      # (a call on a closed handle is refused by the library: HIPNLP_E_INVALID)
            self._addresses = {}
            self._transient = self._transient_hess = None

    def __del__(self):
        # (not while the interpreter is shutting down: the HIP runtime's own exit handlers may already have run, and the OS reclaims
        #  device memory, registrations and streams with the process)
        try:
            if _sys is None or _sys.is_finalizing():
                return
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def _check(self, rc):
        if rc != 0:
            raise HipNlpError(rc, self.lib.hipnlp_last_error(self.h).decode())

    def set_params(self, p):
        p = np.ascontiguousarray(p, dtype=np.float64).reshape(self.batch, self.np)
        self._check(self.lib.hipnlp_set_params(self.h, _dp(p)))
        self.params_generation += 1

    def bounds(self):
        lbx, ubx, lbg, ubg = np.empty(self.n), np.empty(self.n), np.empty(self.m), np.empty(self.m)
        self._check(self.lib.hipnlp_bounds(self.h, _dp(lbx), _dp(ubx), _dp(lbg), _dp(ubg)))
        return lbx, ubx, lbg, ubg

    def simple_rows(self):
        """(is_simple, var_index) over the FULL subject_to list (m_full rows)"""
        a, b = np.zeros(self.m_full, np.int32), np.zeros(self.m_full, np.int32)
        self._check(self.lib.hipnlp_simple_rows(self.h, _ip(a), _ip(b)))
        return a, b

    def lift_map(self, with_bounds=True):
        """(kept_row [m_full], lb_full, ub_full): this handle's row behind every row of the full list (-1: lifted into a bound) and the
        canonical bounds of every full row (None without `with_bounds`; they need the parameters)"""
        kept = np.zeros(self.m_full, np.int32)
        lb = np.empty(self.m_full) if with_bounds else None
        ub = np.empty(self.m_full) if with_bounds else None
        self._check(self.lib.hipnlp_lift_map(self.h, _ip(kept), _dp(lb), _dp(ub)))
        return kept, lb, ub

    def sparsity(self):
        ir, jc = np.zeros(self.nnz, np.int32), np.zeros(self.nnz, np.int32)
        self._check(self.lib.hipnlp_sparsity(self.h, _ip(ir), _ip(jc)))
        return ir, jc

    def row_blocks(self):
        out = []
        for i in range(self.lib.hipnlp_num_row_blocks(self.h)):
            name = C.c_char_p()
            a, b, c, d = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
            self._check(self.lib.hipnlp_row_block(self.h, i, C.byref(name), C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
            out.append((name.value.decode(), a.value, b.value, c.value, d.value))
        return out

    def eval(self, x, new_x=True, want=("f", "grad", "g", "jac"), out=None, nan_ok=False):
        """Host-buffer callback set.  Returns (f[batch], grad[batch,n], g[batch,m], jac[batch,nnz]); only the outputs in `want`
        cross PCIe (the others are None).
        out: (f, grad, g, jac) arrays of a previous call to fill again (what a C caller such as IPOPT does with its own arrays:
        fresh 1 MB numpy arrays cost an mmap and their page faults on every call).
        nan_ok: a non-finite evaluation (HIPNLP_E_NUMERIC) returns the arrays as filled — NaN/Inf included — instead of raising:
        what an NLP driver wants at a line-search trial point (the reference hands CasADi's NaNs to IPOPT, which cuts the step)."""
        if not (type(x) is np.ndarray and x.dtype == np.float64 and x.flags.c_contiguous and x.size == self.batch * self.n):
            x = np.ascontiguousarray(x, dtype=np.float64).reshape(self.batch, self.n)
        if out is not None:
            f, grad, g, jac = out
        else:
            f = np.empty(self.batch) if "f" in want else None
            grad = np.empty((self.batch, self.n)) if "grad" in want else None
            g = np.empty((self.batch, self.m)) if "g" in want else None
            jac = np.empty((self.batch, self.nnz)) if "jac" in want else None
        # (new_x = None: unknown — the library compares x with its staging copy of the previous evaluation)
        # One foreign call on raw addresses; the address of an array is looked up once per array OBJECT (a solver hands in the same
        # arrays iterate after iterate: five `ndarray.ctypes` objects per call were 6 of the 47 us of a 100-knot callback).
        rc = self._eval_raw(self._h_value, self._address(x), -1 if new_x is None else (1 if new_x else 0),
                            self._address(f), self._address(grad), self._address(g), self._address(jac))
        if out is None:
            # Arrays allocated HERE are transient: the caller drops them when it likes, and the allocator hands their addresses out
            # again at once — "the same array twice in a row" for the library, which would page-lock memory that is freed a moment
            # later (hipnlp_set_auto_register is meant for a solver's own arrays, which outlive the solve).  Holding the previous
            # call's arrays until the next one has allocated its own makes two consecutive calls see different addresses.
            self._transient = (f, grad, g, jac)
        if not (nan_ok and rc == -5):
            self._check(rc)
        return f, grad, g, jac

    def _address(self, a):
        """data address of a numpy array, remembered per array object (a weak reference tells a live object from a reused id)"""
        if a is None:
            return None
        ent = self._addresses.get(id(a))
        if ent is not None and ent[0]() is a:
            return ent[1]
        if len(self._addresses) > 256:
            self._addresses.clear()
        addr = a.ctypes.data
        try:
            self._addresses[id(a)] = (weakref.ref(a), addr)
        except TypeError:      # (not weakly referenceable: looked up every time)
            pass
        return addr

    def raw_eval(self):
        """hipnlp_eval as ONE foreign call on raw addresses: `call(x_addr, new_x, f_addr, grad_addr, g_addr, jac_addr) -> rc` (None for an
        output that is not wanted).  What a C binding pays per callback: `eval` above spends several microseconds per call converting
        its numpy arguments, which is a fifth of a 100-knot callback."""
        proto = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)(("hipnlp_eval", self.lib))
        h = self.h.value

        def call(x_addr, new_x, f_addr, grad_addr, g_addr, jac_addr):
            return proto(h, x_addr, new_x, f_addr, grad_addr, g_addr, jac_addr)
        return call

    WANT = {"f": 1, "grad": 2, "g": 4, "jac": 8}

    def eval_pinned(self, x, new_x=True, want=("f", "grad", "g", "jac")):
        """Zero-copy host callback set: (f[batch], grad[batch,n], g[batch,m], jac[batch,nnz]) as numpy VIEWS of the library's pinned
        output block (None for outputs not in `want`); valid until the next evaluation on this handle.  A non-finite evaluation
        raises HipNlpError(-5) with the views left in `self.last_views` (the caller may hand the NaNs on, as IPOPT expects)."""
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(self.batch, self.n)
        mask = 0
        for w in want:
            mask |= self.WANT[w]
        ptrs = [C.POINTER(C.c_double)() for _ in range(4)]
        rc = self.lib.hipnlp_eval_pinned(self.h, _dp(x), 1 if new_x else 0, mask, *[C.byref(q) for q in ptrs])
        shapes = ((self.batch,), (self.batch, self.n), (self.batch, self.m), (self.batch, self.nnz))
        self.last_views = tuple(np.ctypeslib.as_array(q, shape=sh) if q else None for q, sh in zip(ptrs, shapes)) if rc in (0, -5) else None
        self._check(rc)
        return self.last_views

    def set_early_outputs(self, on=True, grad=False):
        """a new evaluation also fills the registered g / jac arrays earlier calls passed (hipnlp_set_early_outputs; see include/hipnlp.h);
        grad=True: grad f too — only for callers whose grad array is scratch of their own (NOT IPOPT's own gradient vector)"""
        self._check(self.lib.hipnlp_set_early_outputs(self.h, (2 if grad else 1) if on else 0))

    def set_prefetch(self, want=("f", "grad", "g")):
        """outputs every new evaluation brings to the host besides the ones its call asks for (hipnlp_set_prefetch)"""
        mask = 0
        for w in want:
            mask |= self.WANT[w]
        self._check(self.lib.hipnlp_set_prefetch(self.h, mask))

    def host_breakdown(self):
        """us of the last host-path evaluation: (x staging copy, enqueue, wait for the GPU, copies into unregistered caller arrays)"""
        us = np.zeros(4)
        self._check(self.lib.hipnlp_host_breakdown(self.h, _dp(us)))
        return us

    def register_outputs(self, arrays):
        """hipnlp_host_register for each numpy array: the kernel then stores straight into them when they are passed as `out`"""
        for a in arrays:
            if a is not None and a.nbytes >= 4096:
                rc = self.lib.hipnlp_host_register(C.c_void_p(a.ctypes.data), C.c_size_t(a.nbytes), None)
                if rc != 0:
                    raise HipNlpError(rc, "hipnlp_host_register failed")

    def unregister_outputs(self, arrays):
        for a in arrays:
            if a is not None and a.nbytes >= 4096:
                self.lib.hipnlp_host_unregister(C.c_void_p(a.ctypes.data))

    def set_auto_register(self, on=True):
        """hipnlp_set_auto_register: output arrays seen twice in a row at one address become direct kernel outputs (on by default)"""
        self._check(self.lib.hipnlp_set_auto_register(self.h, 1 if on else 0))

    def host_stats(self):
        out = (C.c_long * 8)()
        self._check(self.lib.hipnlp_host_stats(self.h, out))
        return dict(zip(("auto_registered", "auto_fallbacks", "auto_ranges", "evaluations", "constant_fills", "constant_refills", "constant_entries", "constant_slices_healed"), list(out)))

    def shards(self):
        """hipnlp_multi_info: [{"device", "knot_begin", "knot_end", "waves"}] of a multi-device handle ([] for a plain one)"""
        n = C.c_int32(64)
        arr = [(C.c_int32 * 64)() for _ in range(4)]
        self._check(self.lib.hipnlp_multi_info(self.h, C.byref(n), *arr))
        return [dict(zip(("device", "knot_begin", "knot_end", "waves"), (int(a[i]) for a in arr))) for i in range(n.value)]

    def set_threads(self, on=True, spin_us=-1.0):
        """hipnlp_multi_set_threads: one launching thread per shard of a multi-device handle (spin_us: how long a worker polls for the
        next job before it sleeps; < 0: unchanged)"""
        self._check(self.lib.hipnlp_multi_set_threads(self.h, 1 if on else 0, float(spin_us)))

    def multi_breakdown(self):
        """us [shards, 2] of the last evaluation of a multi-device handle: (enqueued, seen complete) on the host clock, from the start of the launch loop"""
        out = np.zeros((len(self.shards()), 2))
        self._check(self.lib.hipnlp_multi_breakdown(self.h, _dp(out)))
        return out

    def set_constant_jacobian(self, on=True):
        """hipnlp_set_constant_jacobian: destinations of jac g hold the constant entries, launches store the varying ones only.  Default: on
        for varying-first handles (host and device destinations); a handle in CCS order stores every entry unless this asks for the
        scheme — then its DEVICE destinations (`eval_device`) are filled once and receive the varying entries at their CCS positions."""
        self._check(self.lib.hipnlp_set_constant_jacobian(self.h, 1 if on else 0))
        self.constants_in_place = bool(on)

    def forget_jac_destination(self, ptr=0):
        """hipnlp_forget_jac_destination: the buffer at `ptr` (0: every one the handle remembers) no longer counts as holding the constant
        entries — for callers whose allocator may hand out the address of a freed jac buffer again; returns the records dropped"""
        n = self.lib.hipnlp_forget_jac_destination(self.h, C.c_void_p(ptr or None))
        if n < 0:
            self._check(n)
        return n

    def jac_constant_mask(self):
        """bool [nnz] in the order of `sparsity()`: entries of jac g that do not depend on x"""
        mask = np.zeros(self.nnz, np.uint8)
        self._check(self.lib.hipnlp_jac_constant_mask(self.h, mask.ctypes.data_as(C.POINTER(C.c_ubyte))))
        return mask.astype(bool)

    def set_host_timing(self, on=True):
        self._check(self.lib.hipnlp_set_host_timing(self.h, 1 if on else 0))

    def eval_device(self, x_ptr, f_ptr=0, grad_ptr=0, g_ptr=0, jac_ptr=0, stream=0):
        """Device-pointer variant (ints from tensor.data_ptr()); asynchronous on `stream`."""
        self._check(self.lib.hipnlp_eval_device(self.h, C.c_void_p(x_ptr), C.c_void_p(f_ptr or None), C.c_void_p(grad_ptr or None),
                                                C.c_void_p(g_ptr or None), C.c_void_p(jac_ptr or None), C.c_void_p(stream or None)))

    # ---- exchanges without the constants of jac g (include/hipnlp.h; varying-first handles) ---------------------------------
    def jac_vary_layout(self):
        """hipnlp_jac_vary_layout as a dict: varying entries of a first / interior / last knot block, of the whole horizon, in front of
        this handle's first knot, of this handle's knots"""
        out = (C.c_int64 * 6)()
        self._check(self.lib.hipnlp_jac_vary_layout(self.h, out))
        return dict(zip(("first", "interior", "last", "total", "shard_off", "shard_len"), [int(v) for v in out]))

    def fill_jac_constants(self, jac_ptr, whole_horizon=True, stream=0):
        """the constant entries of jac g into the complete value array [batch][nnz] at jac_ptr (device-visible address)"""
        self._check(self.lib.hipnlp_fill_jac_constants(self.h, C.c_void_p(jac_ptr), 1 if whole_horizon else 0, C.c_void_p(stream or None)))

    def eval_device_vary(self, x_ptr, f_ptr=0, grad_ptr=0, g_ptr=0, jac_vary_ptr=0, stream=0):
        """eval_device with a COMPACT jac destination [batch][jac_vary_layout()["total"]]: the varying runs of the knot blocks only"""
        vp = C.c_void_p
        self._check(self.lib.hipnlp_eval_device_vary(self.h, vp(x_ptr), vp(f_ptr or None), vp(grad_ptr or None), vp(g_ptr or None), vp(jac_vary_ptr or None), vp(stream or None)))

    def eval_device_shard_vary(self, x_ptr, f_ptr, grad_ptr, stage_ptr, jac_vary_ptr, stream=0):
        vp = C.c_void_p
        self._check(self.lib.hipnlp_eval_device_shard_vary(self.h, vp(x_ptr), vp(f_ptr or None), vp(grad_ptr or None), vp(stage_ptr or None), vp(jac_vary_ptr or None), vp(stream or None)))

    def eval_device_peers_vary(self, x_ptr, peer_out_ptr, world, rank, stream=0):
        self._check(self.lib.hipnlp_eval_device_peers_vary(self.h, C.c_void_p(x_ptr), C.c_void_p(peer_out_ptr), int(world), int(rank), C.c_void_p(stream or None)))

    # ---- exact Hessian of the Lagrangian (IPOPT eval_h): lower-triangle triplets, block diagonal by knot ---------------------
    def hess_nnz(self):
        if not hasattr(self, "_hnnz"):
            n = C.c_int64()
            self._check(self.lib.hipnlp_hess_nnz(self.h, C.byref(n)))
            self._hnnz = n.value
        return self._hnnz

    def hess_sparsity(self):
        ir, jc = np.zeros(self.hess_nnz(), np.int32), np.zeros(self.hess_nnz(), np.int32)
        self._check(self.lib.hipnlp_hess_sparsity(self.h, _ip(ir), _ip(jc)))
        return ir, jc

    def eval_hess(self, x, obj_factor, lam, out=None, new_x=True):
        """values [batch][nnz_h] of  obj_factor * hess f + sum_r lam_r hess g_r  at x (obj_factor: scalar or [batch]).
        out: the array of a previous call to fill again (a fresh 1.2 MB numpy array costs its page faults on every call).
        new_x: IPOPT's flag of eval_h — False: x is the x of the previous eval / eval_hess call on this handle (its staged copy is used);
        None: unknown (compared)."""
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(self.batch, self.n)
        lam = np.ascontiguousarray(lam, dtype=np.float64).reshape(self.batch, self.m)
        sig = np.ascontiguousarray(np.broadcast_to(np.asarray(obj_factor, dtype=np.float64), (self.batch,)))
        if out is None:
            out = np.empty((self.batch, self.hess_nnz()))
            self._transient_hess = out     # (see eval: no two consecutive calls on the same freshly allocated address)
        self._check(self.lib.hipnlp_eval_hess_at(self.h, _dp(x), -1 if new_x is None else int(bool(new_x)), _dp(sig), _dp(lam), _dp(out)))
        return out

    def set_hessian_early_run(self, mode):
        """eval_hess into host memory: the run at the start of every knot block leaves early — True / False, or None: the handle decides from
        its own first calls (the default; include/hipnlp.h)"""
        self._check(self.lib.hipnlp_set_hessian_early_run(self.h, -1 if mode is None else int(bool(mode))))

    def hessian_early_run(self):
        """{"mode": None | bool, "in_use": None (not decided yet) | bool, "us_off", "us_on": the medians a MEASURED decision rests on (0: decided
        without a clock), "why": how the choice was made}"""
        mode, chosen, off, on = C.c_int(), C.c_int(), C.c_double(), C.c_double()
        self._check(self.lib.hipnlp_get_hessian_early_run(self.h, C.byref(mode), C.byref(chosen), C.byref(off), C.byref(on)))
        tri = lambda v: None if v < 0 else bool(v)  # noqa: E731
        return {"mode": tri(mode.value), "in_use": tri(chosen.value), "us_off": off.value, "us_on": on.value,
                "why": self.lib.hipnlp_hessian_early_run_reason(self.h).decode()}

    def eval_hess_device(self, x_ptr, obj_factor_ptr, lam_ptr, hess_ptr, stream=None):
        vp = C.c_void_p
        self._check(self.lib.hipnlp_eval_hess_device(self.h, vp(x_ptr), vp(obj_factor_ptr), vp(lam_ptr), vp(hess_ptr), vp(stream or None)))

    def cost_terms(self):
        out = np.empty((self.batch, _abi.NCOST_TERMS))
        self._check(self.lib.hipnlp_cost_terms(self.h, _dp(out)))
        names = [self.lib.hipnlp_cost_term_name(i).decode() for i in range(_abi.NCOST_TERMS)]
        return names, out

    def last_kernel_ms(self):
        ms = C.c_float()
        self._check(self.lib.hipnlp_last_kernel_ms(self.h, C.byref(ms)))
        return ms.value

    def profile_begin(self, max_launches, stride=1):
        """Arm HIP-event timing of every `stride`-th device-path launch (at most max_launches samples)."""
        self._check(self.lib.hipnlp_profile_begin(self.h, int(max_launches), int(stride)))

    def profile_begin_runs(self, max_runs, run_len):
        """HIP events around RUNS of run_len consecutive launches (see include/hipnlp.h)"""
        self._check(self.lib.hipnlp_profile_begin_runs(self.h, int(max_runs), int(run_len)))

    def profile_end(self):
        """(mean knot-kernel ms, mean launch ms incl. the cost reduction kernel, launches)"""
        a, b, n = C.c_double(), C.c_double(), C.c_int()
        self._check(self.lib.hipnlp_profile_end(self.h, C.byref(a), C.byref(b), C.byref(n)))
        return a.value, b.value, n.value

    def kernels_per_eval(self):
        """1: latency variant (cost summed inside the knot kernel), 2: throughput variant (knot kernel + reduction kernel)"""
        return int(self.lib.hipnlp_kernels_per_eval(self.h))

    def eval_device_shard(self, x_ptr, f_ptr, grad_ptr, g_stage_ptr, jac_ptr, stream=0):
        self._check(self.lib.hipnlp_eval_device_shard(self.h, C.c_void_p(x_ptr), C.c_void_p(f_ptr or None), C.c_void_p(grad_ptr or None),
                                                      C.c_void_p(g_stage_ptr or None), C.c_void_p(jac_ptr or None), C.c_void_p(stream or None)))

    def eval_device_peers(self, x_ptr, peer_out_ptr, world, rank, stream=0):
        """hipnlp_eval_device_peers: the shard's outputs at their final positions in every rank's [grad | jac | g | f partials | f] buffer"""
        self._check(self.lib.hipnlp_eval_device_peers(self.h, C.c_void_p(x_ptr), C.c_void_p(peer_out_ptr), int(world), int(rank), C.c_void_p(stream or None)))

    def stage_rows(self, k):
        rows = np.zeros(G_STAGE, np.int32)
        self._check(self.lib.hipnlp_stage_rows(self.h, int(k), _ip(rows)))
        return rows


class HipPose:
    """One pose-finder handle (hipnlp_pose_*): `batch` independent static poses per launch on one HIP device."""

    def __init__(self, settings, model, batch=1, device=0, desc=None):
        self.lib = load_library()
        if desc is None:
            self.desc = _abi.PoseDescC()
            self.desc.settings = settings.to_c()
            self.desc.model = model.to_c()
            self.desc.batch = int(batch)
            self.desc.device = int(device)
        else:   # a ready hipnlp_pose_desc (from_reference.pose_from_reference, a committed fixture)
            self.desc = desc
            batch = int(desc.batch)
        h = C.c_void_p()
        rc = self.lib.hipnlp_pose_create(C.byref(self.desc), C.byref(h))
        if rc != 0:
            raise HipNlpError(rc, self.lib.hipnlp_pose_last_error(None).decode())
        self.h = h
        d = _abi.PoseDimsC()
        self._check(self.lib.hipnlp_pose_get_dims(self.h, C.byref(d)))
        self.batch = int(batch)
        self.n, self.m, self.nnz, self.np = d.n, d.m, d.nnz, d.np

    @classmethod
    def from_desc(cls, desc):
        """a pose handle from a hipnlp_pose_desc structure or its raw bytes"""
        if not isinstance(desc, _abi.PoseDescC):
            desc = _abi.PoseDescC.from_buffer_copy(bytes(desc))
        return cls(None, None, desc=desc)

    def close(self):
        if getattr(self, "h", None):
            self.lib.hipnlp_pose_destroy(self.h)
            self.h = None

    def __del__(self):
        # (not while the interpreter is shutting down: the HIP runtime's own exit handlers may already have run, and the OS reclaims
        #  device memory, registrations and streams with the process)
        try:
            if _sys is None or _sys.is_finalizing():
                return
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def _check(self, rc):
        if rc != 0:
            raise HipNlpError(rc, self.lib.hipnlp_pose_last_error(self.h).decode())

    def set_params(self, p):
        p = np.ascontiguousarray(p, dtype=np.float64).reshape(self.batch, self.np)
        self._check(self.lib.hipnlp_pose_set_params(self.h, _dp(p)))

    def bounds(self):
        lb, ub = np.zeros((self.batch, self.m)), np.zeros((self.batch, self.m))
        self._check(self.lib.hipnlp_pose_bounds(self.h, _dp(lb), _dp(ub)))
        return lb, ub

    def sparsity(self):
        ir, jc = np.zeros(self.nnz, np.int32), np.zeros(self.nnz, np.int32)
        self._check(self.lib.hipnlp_pose_sparsity(self.h, _ip(ir), _ip(jc)))
        return ir, jc

    def eval(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(self.batch, self.n)
        f, grad = np.zeros(self.batch), np.zeros((self.batch, self.n))
        g, jac = np.zeros((self.batch, self.m)), np.zeros((self.batch, self.nnz))
        self._check(self.lib.hipnlp_pose_eval(self.h, _dp(x), _dp(f), _dp(grad), _dp(g), _dp(jac)))
        return f, grad, g, jac

    def eval_device(self, x_ptr, f_ptr, grad_ptr, g_ptr, jac_ptr, stream=None):
        """Device-pointer variant (ints from tensor.data_ptr()); asynchronous on `stream`.  (Every pointer goes through c_void_p:
        a bare Python int would be passed as a 32-bit C int.)"""
        vp = C.c_void_p
        self._check(self.lib.hipnlp_pose_eval_device(self.h, vp(x_ptr), vp(f_ptr or None), vp(grad_ptr or None), vp(g_ptr or None),
                                                     vp(jac_ptr or None), vp(stream or None)))

    # ---- exact Hessian of the Lagrangian (IPOPT eval_h): lower triangle, column major ------------------------------------
    def hess_sparsity(self):
        n = C.c_int32()
        self._check(self.lib.hipnlp_pose_hess_nnz(self.h, C.byref(n)))
        ir, jc = np.zeros(n.value, np.int32), np.zeros(n.value, np.int32)
        self._check(self.lib.hipnlp_pose_hess_sparsity(self.h, _ip(ir), _ip(jc)))
        return ir, jc

    def eval_hess(self, x, obj_factor, lam):
        """values [batch][nnz_h] of  obj_factor * hess f + sum_r lam_r hess g_r  at x (obj_factor: scalar or [batch])"""
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(self.batch, self.n)
        lam = np.ascontiguousarray(lam, dtype=np.float64).reshape(self.batch, self.m)
        sig = np.ascontiguousarray(np.broadcast_to(np.asarray(obj_factor, dtype=np.float64), (self.batch,)))
        if not hasattr(self, "_hnnz"):
            n = C.c_int32()
            self._check(self.lib.hipnlp_pose_hess_nnz(self.h, C.byref(n)))
            self._hnnz = n.value
        out = np.zeros((self.batch, self._hnnz))
        self._check(self.lib.hipnlp_pose_eval_hess(self.h, _dp(x), _dp(sig), _dp(lam), _dp(out)))
        return out

    def eval_hess_device(self, x_ptr, obj_factor_ptr, lam_ptr, hess_ptr, stream=None):
        vp = C.c_void_p
        self._check(self.lib.hipnlp_pose_eval_hess_device(self.h, vp(x_ptr), vp(obj_factor_ptr), vp(lam_ptr), vp(hess_ptr), vp(stream or None)))

    def cost_terms(self):
        names = [self.lib.hipnlp_pose_cost_term_name(i).decode() for i in range(_abi.POSE_NCOST_TERMS)]
        v = np.zeros((self.batch, _abi.POSE_NCOST_TERMS))
        self._check(self.lib.hipnlp_pose_cost_terms(self.h, _dp(v)))
        return names, v

    def row_blocks(self):
        out = []
        for i in range(self.lib.hipnlp_pose_num_row_blocks(self.h)):
            name = C.c_char_p()
            a, b = C.c_int32(), C.c_int32()
            self._check(self.lib.hipnlp_pose_row_block(self.h, i, C.byref(name), C.byref(a), C.byref(b)))
            out.append((name.value.decode(), a.value, b.value))
        return out

    def set_host_timing(self, on=True):
        """host-buffer calls bracket their launch with HIP events (off by default: the records cost ~3 us of a 25 us call)"""
        self.lib.hipnlp_pose_set_host_timing.argtypes = [C.c_void_p, C.c_int]
        self._check(self.lib.hipnlp_pose_set_host_timing(self.h, 1 if on else 0))

    def last_kernel_ms(self):
        ms = C.c_float()
        self._check(self.lib.hipnlp_pose_last_kernel_ms(self.h, C.byref(ms)))
        return ms.value
