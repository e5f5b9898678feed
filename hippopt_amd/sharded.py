"""Knot-interval data parallelism: contiguous knot shards, one process per GPU, ONE
all-gather (RCCL over xGMI via torch.distributed backend "nccl"; "gloo" in the CPU tests) of a fused
per-rank output buffer, then one index-select that reassembles [grad f | jac values | g] in the
reference's order.  The reference has no distributed code at all; this is the
build's addition for horizons sharded across the GPUs of one node.

Fused shard buffer of rank r (all ranks use the same padded length so the all-gather is regular):
    [ f partial (1) | grad shard (glen_max) | jac shard (jlen_max) | g staging (nk_max * G_STAGE) ]
"""
import math

import numpy as np
import torch
import torch.distributed as dist

G_STAGE = 550


def knot_range(horizon, world, rank):
    """GPU r of R owns knots [r*ceil(N/R), (r+1)*ceil(N/R))  (SURVEY §8e)."""
    per = int(math.ceil(horizon / world))
    return min(rank * per, horizon), min((rank + 1) * per, horizon)


class ShardedCallback:
    """compute_shard(x, f_view, grad_view, jac_view, stage_view) fills the rank's views (any backend)."""

    def __init__(self, horizon, n, m, nnz, shard_info, compute_shard, device, group=None):
        """shard_info: dict(glen, jlen, nk, stage_rows [nk, G_STAGE] int32 global rows or -1) of THIS rank."""
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.n, self.m, self.nnz, self.horizon = n, m, nnz, horizon
        self.compute_shard = compute_shard
        self.device = device
        infos = [None] * self.world
        if self.world > 1:
            dist.all_gather_object(infos, shard_info, group=group)
        else:
            infos = [shard_info]
        self.infos = infos
        self.glen_max = max(i["glen"] for i in infos)
        self.jlen_max = max(i["jlen"] for i in infos)
        self.nk_max = max(i["nk"] for i in infos)
        self.o_grad = 1
        self.o_jac = 1 + self.glen_max
        self.o_stage = self.o_jac + self.jlen_max
        self.shard_len = self.o_stage + self.nk_max * G_STAGE
        self.buf = torch.zeros(self.shard_len, dtype=torch.float64, device=device)
        self.all = torch.zeros(self.world * self.shard_len, dtype=torch.float64, device=device)
        me = infos[self.rank]
        self.views = (
            self.buf[0:1],
            self.buf[self.o_grad:self.o_grad + me["glen"]],
            self.buf[self.o_jac:self.o_jac + me["jlen"]],
            self.buf[self.o_stage:self.o_stage + me["nk"] * G_STAGE],
        )
        # index of every entry of [grad | jac | g] inside the gathered buffer
        src = np.full(n + nnz + m, -1, dtype=np.int64)
        go, jo = 0, 0
        for r, inf in enumerate(infos):
            base = r * self.shard_len
            src[go:go + inf["glen"]] = base + self.o_grad + np.arange(inf["glen"])
            go += inf["glen"]
            src[n + jo:n + jo + inf["jlen"]] = base + self.o_jac + np.arange(inf["jlen"])
            jo += inf["jlen"]
            rows = np.asarray(inf["stage_rows"], dtype=np.int64).reshape(-1)
            valid = np.nonzero(rows >= 0)[0]
            src[n + nnz + rows[valid]] = base + self.o_stage + valid
        if go != n or jo != nnz or (src < 0).any():
            raise ValueError("shards do not tile the problem: grad %d/%d jac %d/%d missing %d" % (go, n, jo, nnz, int((src < 0).sum())))
        self.src = torch.from_numpy(src).to(device)
        self.f_src = torch.arange(self.world, device=device) * self.shard_len
        self.out = torch.empty(n + nnz + m + 1, dtype=torch.float64, device=device)   # [grad | jac | g | f]
        self._lib = None
        if torch.device(device).type == "cuda":   # one HIP launch for the whole reassembly (hipnlp_reassemble); CPU tests: torch ops
            import ctypes as C
            from .hipnlp import load_library
            self._lib = load_library()
            self._lib.hipnlp_reassemble.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]

    def __call__(self, x):
        """One callback set for the whole horizon.  Returns (f, grad, jac, g) views of the reassembled buffer."""
        self.compute_shard(x, *self.views)
        if self.world > 1:
            dist.all_gather_into_tensor(self.all, self.buf, group=self.group)
            gathered = self.all
        else:
            gathered = self.buf
        tot = self.n + self.nnz + self.m
        if self._lib is not None:
            rc = self._lib.hipnlp_reassemble(gathered.data_ptr(), self.src.data_ptr(), self.out.data_ptr(), tot, self.world, self.shard_len,
                                             self.out.data_ptr() + 8 * tot, torch.cuda.current_stream().cuda_stream)
            if rc != 0:
                raise RuntimeError("hipnlp_reassemble failed (%d)" % rc)
            f = self.out[tot]
        else:
            torch.index_select(gathered, 0, self.src, out=self.out[:tot])
            f = gathered.index_select(0, self.f_src).sum()
        return f, self.out[:self.n], self.out[self.n:self.n + self.nnz], self.out[self.n + self.nnz:tot]


def hip_shard_backend(engine):
    """compute_shard backed by the HIP engine handle that owns this rank's knots."""
    def compute(x, f_view, grad_view, jac_view, stage_view):
        engine.eval_device_shard(x.data_ptr(), f_view.data_ptr(), grad_view.data_ptr(), stage_view.data_ptr(), jac_view.data_ptr(),
                                 stream=torch.cuda.current_stream().cuda_stream)
    return compute


def hip_shard_info(engine, knot_begin, knot_end):
    d = engine.dims
    rows = np.stack([engine.stage_rows(k) for k in range(knot_begin, knot_end)])
    return {"glen": int(d.shard_grad), "jlen": int(d.shard_nnz), "nk": knot_end - knot_begin, "stage_rows": rows}
