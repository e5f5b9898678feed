"""Knot-interval data parallelism: contiguous knot shards, one process per GPU.  The reference has no
distributed code at all; this is the build's addition for horizons sharded across the GPUs of one node.

Two ways to hand the shards' outputs to the consumer of the callback:

* `ShardedCallback` — north_star's exchange: ONE all-gather (RCCL over xGMI via torch.distributed backend "nccl"; "gloo" in the
  CPU tests) of a fused per-rank output buffer, then ONE launch (hipnlp_reassemble) that writes [grad f | jac values | g] in the
  reference's order on every rank.  Fused shard buffer of rank r (all ranks use the same padded length, the all-gather is regular):
      [ f partial (1) | grad shard (glen_max) | jac shard (jlen_max) | g staging (nk_max * G_STAGE) ]
* `PeerExchange` — the same result as `ShardedCallback` without a collective and without a reassembly pass: every rank pushes its
  fused shard buffer, entry by entry at its final position, into the output buffer of EVERY rank with plain stores over xGMI
  (buffers shared through HIP IPC handles), then raises a flag on every rank; torch.distributed only moves the 64-byte handles
  once, at set-up.
  Without the constants of jac g (`const_mask` / `const_fill`; shard handles created with HIPNLP_FLAG_JAC_VARYING_FIRST): 43 % of the
  Jacobian pattern never changes between callbacks (the +-1, -dt/2, mass entries of the linear rows the transcription emits,
  base/multiple_shooting_solver.py:713-742).  The fused shard buffer then carries the VARYING RUN of every knot block only, the
  reassembled buffer of every rank (rank 0's alone for the gathers to the root) is filled with the constants once per parameter set
  (`refresh_constants()`, again after every set_params), and the reassembly writes the entries the shards sent and nothing else:
  0.68 of the bytes on the links, the exchange being the bound of this path.
* `HostSink` — SURVEY §5's alternative for a CPU-side IPOPT: no collective at all.  Every rank's knot kernel stores its shard of
  g / jac / grad f straight into ONE shared, pinned host buffer (a POSIX shared-memory segment every rank maps and registers with the
  HIP runtime), already in the reference's order; rank partial costs land in f_parts[rank] and are summed in rank order.
"""
import mmap
import os

import numpy as np
import torch
import torch.distributed as dist

G_STAGE = 550


def knot_range(horizon, world, rank):
    """Contiguous, balanced, never empty: the first (horizon mod world) ranks own one knot more (SURVEY §8e partitioning;
    defects of interval k -> k+1 belong to the owner of knot k+1, base/multiple_shooting_solver.py:728)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    if world > horizon:
        raise ValueError(f"{world} ranks cannot share a horizon of {horizon} knots (every rank needs at least one knot)")
    q, r = divmod(horizon, world)
    begin = rank * q + min(rank, r)
    return begin, begin + q + (1 if rank < r else 0)


class XFeed:
    """The INPUT half of a sharded callback.  The reference has one caller of the callbacks — IPOPT, inside the process that runs
    `self._solver.solve()` (/root/reference/src/hippopt/base/opti_solver.py:479) — so with one process per GPU every new x exists in
    rank 0's HOST memory and nowhere else.  `feed(x_host)` brings it to every rank's device: rank 0 copies it into its device buffer
    (from pinned memory, asynchronously on the current stream) and broadcasts that buffer (RCCL broadcast over xGMI through
    torch.distributed backend "nccl"; "gloo" in the CPU tests and in rehearsals with several ranks on one card).  The whole x is
    sent — 1.5 KB per knot, a tenth of what the outputs move back — rather than each rank's knots + one-knot halo + the horizon ends
    (hipnlp_multi_plan lists those ranges): one regular collective instead of `world` irregular sends.
    Returns the device tensor every rank evaluates from; valid until the next feed."""

    def __init__(self, n, device, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.device = torch.device(device)
        self.n = int(n)
        self.x = torch.zeros(self.n, dtype=torch.float64, device=self.device)
        self._host = None   # (gloo moves host tensors only: the rehearsal's staging)

    def stage(self, x):
        """a pinned host copy of x (what a caller that feeds the same arrays again and again keeps): rank 0 only"""
        t = torch.as_tensor(np.ascontiguousarray(x, dtype=np.float64).reshape(-1))
        if t.numel() != self.n:
            raise ValueError("x has %d entries, the problem has %d variables" % (t.numel(), self.n))
        return t.pin_memory() if self.device.type == "cuda" else t.clone()

    def feed(self, x_host=None):
        """x_host: rank 0's host tensor (None on the other ranks).  A collective: every rank calls it."""
        if self.rank == 0:
            if x_host is None:
                raise ValueError("rank 0 feeds x")
            self.x.copy_(x_host.reshape(-1), non_blocking=True)
        if self.world > 1:
            if self.device.type == "cuda" and dist.get_backend(self.group) == "gloo":
                if self._host is None:
                    self._host = torch.empty(self.n, dtype=torch.float64)
                if self.rank == 0:
                    torch.cuda.current_stream(self.device).synchronize()
                    self._host.copy_(self.x)
                dist.broadcast(self._host, 0, group=self.group)
                if self.rank != 0:
                    self.x.copy_(self._host)
            else:
                dist.broadcast(self.x, 0, group=self.group)
        return self.x


class XDeal:
    """XFeed for trajectories dealt over the ranks (BatchDealtCallback, BASELINE config 5's batched initial guesses): the [batch][n]
    array of all guesses exists in rank 0's host memory — where the NLP drivers of the guesses live; `deal(x_host)` copies it to rank
    0's device and scatters every rank ITS trajectories [b0, b1) (dist.scatter: RCCL over xGMI; gloo on the CPU and in rehearsals).
    Returns the rank's [b1 - b0][n] device tensor, valid until the next deal."""

    def __init__(self, batch, n, device, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.device = torch.device(device)
        self.batch, self.n = int(batch), int(n)
        self.b0, self.b1 = batch_range(self.batch, self.world, self.rank)
        self.full = torch.zeros(self.batch, self.n, dtype=torch.float64, device=self.device) if self.rank == 0 else None
        self.local = torch.zeros(self.b1 - self.b0, self.n, dtype=torch.float64, device=self.device)

    def stage(self, x_all):
        t = torch.as_tensor(np.ascontiguousarray(x_all, dtype=np.float64)).reshape(self.batch, self.n)
        return t.pin_memory() if self.device.type == "cuda" else t.clone()

    def deal(self, x_host=None):
        """x_host: rank 0's [batch][n] host tensor (None elsewhere).  A collective: every rank calls it."""
        if self.rank == 0:
            if x_host is None:
                raise ValueError("rank 0 deals x")
            self.full.copy_(x_host, non_blocking=True)
        if self.world == 1:
            self.local.copy_(self.full)
            return self.local
        per = self.b1 - self.b0
        if self.device.type == "cuda" and dist.get_backend(self.group) == "gloo":     # (rehearsal: gloo moves host tensors)
            host = torch.empty(per, self.n, dtype=torch.float64)
            parts = None
            if self.rank == 0:
                torch.cuda.current_stream(self.device).synchronize()
                parts = [c.contiguous() for c in self.full.cpu().split(per)]
            dist.scatter(host, parts, src=0, group=self.group)
            self.local.copy_(host)
        else:
            dist.scatter(self.local, list(self.full.split(per)) if self.rank == 0 else None, src=0, group=self.group)
        return self.local


class ShardedCallback:
    """compute_shard(x, f_view, grad_view, jac_view, stage_view, stream) fills the rank's views (any backend).

    Stream discipline on the GPU: shard evaluation, all-gather and reassembly run on ONE dedicated (non-default) torch stream whose
    handle is passed to both library calls — the library would map a NULL stream to the handle's own non-blocking stream for the
    evaluation but to the legacy default stream for hipnlp_reassemble, and nothing orders those two.  The dedicated stream waits
    for the caller's current stream first (x is ready) and the caller's stream waits for it at the end (the returned views are
    ordered for whoever uses them next, and the next call cannot overwrite buffers still being read)."""

    def __init__(self, horizon, n, m, nnz, shard_info, compute_shard, device, group=None, const_mask=None, const_fill=None, param_gen=None):
        """shard_info: dict(glen, jlen, nk, stage_rows [nk, G_STAGE] int32 global rows or -1) of THIS rank.
        param_gen() -> the generation of the engine's parameters (HipNlp.params_generation; hip_constants supplies it): the constants the
        receiving buffers hold are those of ONE parameter set — every call compares and refreshes by itself when set_params has run
        since (a missed refresh_constants() would hand out a Jacobian with the old dt / mass entries, silently).
        const_mask (bool [nnz], the same on every rank: entries of jac g that do not depend on x) + const_fill(jac_view, stream_handle)
        (writes the constant entries of the WHOLE horizon into a [nnz] view of this rank's memory): the shards exchange the varying
        entries only — shard_info then carries "jvary", the varying entries of the rank's jac range, and compute_shard fills a jac view
        of that length (the varying runs of the rank's knot blocks behind one another, in pattern order)."""
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.n, self.m, self.nnz, self.horizon = n, m, nnz, horizon
        self.compute_shard = compute_shard
        self.device = torch.device(device)
        infos = [None] * self.world
        if self.world > 1:
            dist.all_gather_object(infos, shard_info, group=group)
        else:
            infos = [shard_info]
        self.infos = infos
        self.compact = const_mask is not None
        self.const_fill = const_fill
        self.param_gen = param_gen
        self._filled_gen = None          # generation of the parameters whose constants self.out holds
        self._gen_listeners = []         # (PeerExchange: buffers of its own that hold the same constants)
        if self.compact:
            if const_fill is None or any("jvary" not in i for i in infos):
                raise ValueError("const_mask needs const_fill and shard_info['jvary'] on every rank")
            const_mask = np.asarray(const_mask, dtype=bool)
            if const_mask.shape != (nnz,):
                raise ValueError("const_mask must have one flag per entry of the pattern")
        jkey = "jvary" if self.compact else "jlen"
        self.glen_max = max(i["glen"] for i in infos)
        self.jlen_max = max(i[jkey] for i in infos)
        self.nk_max = max(i["nk"] for i in infos)
        # staging of g: G_STAGE native slots per knot (stage_rows: -1 where a knot owns no row) — or, "slen" given, the rows a rank's knots
        # own behind one another (a compact exchange: stage_rows then lists them, all >= 0)
        self.slen_max = max(i.get("slen", i["nk"] * G_STAGE) for i in infos)
        self.o_grad = 1
        self.o_jac = 1 + self.glen_max
        self.o_stage = self.o_jac + self.jlen_max
        self.shard_len = self.o_stage + self.slen_max
        self.buf = torch.zeros(self.shard_len, dtype=torch.float64, device=device)
        self.all = torch.zeros(self.world * self.shard_len, dtype=torch.float64, device=device)
        me = infos[self.rank]
        self.views = (
            self.buf[0:1],
            self.buf[self.o_grad:self.o_grad + me["glen"]],
            self.buf[self.o_jac:self.o_jac + me[jkey]],
            self.buf[self.o_stage:self.o_stage + me.get("slen", me["nk"] * G_STAGE)],
        )
        # index of every entry of [grad | jac | g] inside the gathered buffer; every slot is written exactly once
        src = np.full(n + nnz + m, -1, dtype=np.int64)
        hits = np.zeros(m, dtype=np.int64)
        go, jo = 0, 0
        for r, inf in enumerate(infos):
            base = r * self.shard_len
            src[go:go + inf["glen"]] = base + self.o_grad + np.arange(inf["glen"])
            go += inf["glen"]
            if self.compact:   # the varying entries of the rank's range, in pattern order; the constants have no source
                vary = np.nonzero(~const_mask[jo:jo + inf["jlen"]])[0]
                if vary.size != inf["jvary"]:
                    raise ValueError("rank %d: %d varying entries in its jac range, shard_info says %d" % (r, vary.size, inf["jvary"]))
                src[n + jo + vary] = base + self.o_jac + np.arange(vary.size)
            else:
                src[n + jo:n + jo + inf["jlen"]] = base + self.o_jac + np.arange(inf["jlen"])
            jo += inf["jlen"]
            rows = np.asarray(inf["stage_rows"], dtype=np.int64).reshape(-1)
            valid = np.nonzero(rows >= 0)[0]
            np.add.at(hits, rows[valid], 1)
            src[n + nnz + rows[valid]] = base + self.o_stage + valid
        unsourced = src < 0
        if self.compact:
            unsourced[n:n + nnz] &= ~const_mask
        if go != n or jo != nnz or unsourced.any() or (hits != 1).any():
            raise ValueError("shards do not tile the problem: grad %d/%d jac %d/%d rows missing %d, rows written twice %d"
                             % (go, n, jo, nnz, int((hits == 0).sum()), int((hits > 1).sum())))
        self.src = torch.from_numpy(src).to(device)          # (-1 at the constant entries of a compact exchange)
        if self.compact:
            dst_c = np.nonzero(src >= 0)[0]
            self.dst_c = torch.from_numpy(dst_c).to(device)
            self.src_c = torch.from_numpy(src[dst_c]).to(device)
        self.f_src = torch.arange(self.world, device=device) * self.shard_len
        self.out = torch.zeros(n + nnz + m + 1, dtype=torch.float64, device=device)   # [grad | jac | g | f]
        self._lib = None
        self.stream = None
        if self.device.type == "cuda":   # one HIP launch for the whole reassembly (hipnlp_reassemble); CPU tests: torch ops
            import ctypes as C
            from .hipnlp import load_library
            self._lib = load_library()
            self._lib.hipnlp_reassemble.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]
            self._lib.hipnlp_reassemble_scatter.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]
            self.stream = torch.cuda.Stream(device=self.device)
            if not self.stream.cuda_stream:
                raise RuntimeError("expected a non-default HIP stream")
        self.refresh_constants()

    def refresh_constants(self):
        """(compact exchange) the constant entries of jac g into this rank's reassembled buffer: at set-up, and again after every change of
        the parameters (dt and the mass are what the constants hold).  Ordered on the callback's stream."""
        if not self.compact:
            return
        self._filled_gen = self.param_gen() if self.param_gen is not None else None
        view = self.out[self.n:self.n + self.nnz]
        if self.stream is None:
            self.const_fill(view, 0)
            return
        cur = torch.cuda.current_stream(self.device)
        if cur.cuda_stream != self.stream.cuda_stream:
            self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            self.const_fill(view, self.stream.cuda_stream)
        if cur.cuda_stream != self.stream.cuda_stream:
            cur.wait_stream(self.stream)

    def _fresh(self):
        """the constants in the receiving buffers belong to the engine's CURRENT parameters: refreshed here when set_params has run
        since they were filled (collective-free: every rank fills its own memory, and every rank's set_params bumps its own counter)"""
        if self.compact and self.param_gen is not None and self.param_gen() != self._filled_gen:
            self.refresh_constants()
            for listener in self._gen_listeners:
                listener()

    def bytes_sent_per_step(self):
        """bytes of this rank's fused shard buffer that reach every other rank in one all-gather"""
        return 8 * self.shard_len * (self.world - 1)

    def _reassemble(self, gathered, stream_handle):
        tot = self.n + self.nnz + self.m
        if self._lib is not None:
            if self.compact:
                rc = self._lib.hipnlp_reassemble_scatter(gathered.data_ptr(), self.src_c.data_ptr(), self.dst_c.data_ptr(), self.out.data_ptr(), self.src_c.numel(),
                                                         self.world, self.shard_len, self.out.data_ptr() + 8 * tot, stream_handle)
            else:
                rc = self._lib.hipnlp_reassemble(gathered.data_ptr(), self.src.data_ptr(), self.out.data_ptr(), tot, self.world, self.shard_len,
                                                 self.out.data_ptr() + 8 * tot, stream_handle)
            if rc != 0:
                raise RuntimeError("hipnlp_reassemble failed (%d)" % rc)
            f = self.out[tot]
        else:
            if self.compact:
                self.out[:tot].index_copy_(0, self.dst_c, gathered.index_select(0, self.src_c))
            else:
                torch.index_select(gathered, 0, self.src, out=self.out[:tot])
            f = gathered.index_select(0, self.f_src).sum()
        return f, self.out[:self.n], self.out[self.n:self.n + self.nnz], self.out[self.n + self.nnz:tot]

    def _run(self, x, stream_handle):
        self.compute_shard(x, *self.views, stream_handle)
        if self.world > 1:
            if self.device.type == "cuda" and dist.get_backend(self.group) == "gloo":
                # rehearsal of the plumbing with several ranks on one GPU (RCCL refuses that; gloo gathers host tensors only)
                torch.cuda.current_stream(self.device).synchronize()
                host = torch.empty(self.world * self.shard_len, dtype=torch.float64)
                dist.all_gather_into_tensor(host, self.buf.cpu(), group=self.group)
                self.all.copy_(host)
            else:
                dist.all_gather_into_tensor(self.all, self.buf, group=self.group)
            gathered = self.all
        else:
            gathered = self.buf
        return self._reassemble(gathered, stream_handle)

    def __call__(self, x):
        """One callback set for the whole horizon.  Returns (f, grad, jac, g) views of the reassembled buffer."""
        self._fresh()
        if self.stream is None:
            return self._run(x, 0)
        cur = torch.cuda.current_stream(self.device)
        if cur.cuda_stream == self.stream.cuda_stream:   # the caller already works on the callback's stream (`with
            return self._run(x, self.stream.cuda_stream)  # torch.cuda.stream(cb.stream)`): nothing to order, no event traffic
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            out = self._run(x, self.stream.cuda_stream)
        cur.wait_stream(self.stream)
        return out

    def to_root(self, x):
        """gather_to_root by the collective: ONE consumer (IPOPT lives in rank 0's process), so only rank 0 receives — `dist.gather`
        of the fused shard buffers (RCCL: one send per rank, 1 / world of the all-gather's bytes per link) and rank 0's one-launch
        reassembly.  Returns (f, grad, jac, g) on rank 0, (None, None, None, None) on the others."""
        self._fresh()

        def run(stream_handle):
            self.compute_shard(x, *self.views, stream_handle)
            if self.world > 1:
                chunks = list(self.all.view(self.world, self.shard_len).unbind(0)) if self.rank == 0 else None
                if self.device.type == "cuda" and dist.get_backend(self.group) == "gloo":   # (rehearsal: several ranks on one GPU)
                    torch.cuda.current_stream(self.device).synchronize()
                    host = [torch.empty(self.shard_len, dtype=torch.float64) for _ in range(self.world)] if self.rank == 0 else None
                    dist.gather(self.buf.cpu(), host, dst=0, group=self.group)
                    if self.rank == 0:
                        self.all.copy_(torch.cat(host))
                else:
                    dist.gather(self.buf, chunks, dst=0, group=self.group)
                if self.rank != 0:
                    return None, None, None, None
                gathered = self.all
            else:
                gathered = self.buf
            return self._reassemble(gathered, stream_handle)
        if self.stream is None:
            return run(0)
        cur = torch.cuda.current_stream(self.device)
        if cur.cuda_stream == self.stream.cuda_stream:
            return run(self.stream.cuda_stream)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            out = run(self.stream.cuda_stream)
        cur.wait_stream(self.stream)
        return out

    def shard_only(self, x):
        """The rank's shard evaluated into its fused buffer, nothing exchanged (bench.py: outputs left shard-resident)."""
        if self.stream is None:
            self.compute_shard(x, *self.views, 0)
            return
        cur = torch.cuda.current_stream(self.device)
        if cur.cuda_stream == self.stream.cuda_stream:
            self.compute_shard(x, *self.views, self.stream.cuda_stream)
            return
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            self.compute_shard(x, *self.views, self.stream.cuda_stream)
        cur.wait_stream(self.stream)


class PeerExchange:
    """The reassembled [grad | jac | g | f] of `ShardedCallback` on every rank by peer stores instead of all-gather + reassembly
    (include/hipnlp.h, "the same exchange WITHOUT a collective").  Built on a ShardedCallback (same shard buffer, same tiling):

        px = PeerExchange(cb)
        f, grad, jac, g = px(x)          # views of this rank's output buffer of the step's parity

    With `engine` (the rank's HipNlp shard handle; shards of at most 256 knots) the push is folded into the evaluation
    (hipnlp_eval_device_peers): the knot kernel itself stores the shard's entries at their final positions in every rank's buffer —
    the link transfers overlap with the knot programs still running, and a step is three launches (evaluate, signal, wait)
    instead of four plus the staging of g.

    Two output buffers alternate between steps: a rank starts pushing step i + 1 once every rank has signalled step i, i.e. has
    finished its own push of step i — at which point a slow rank may still be READING step i, so step i + 1 goes to the other
    buffer (and step i + 2 cannot start before everyone has signalled i + 1, which comes behind their use of step i in stream order).
    The caller orders its consumption of the views before its next call on the same stream, as with ShardedCallback."""

    def __init__(self, cb, engine=None, root_only=False, handshake_timeout_s=5.0):
        """root_only: `gather_to_root` — only rank 0 (the process IPOPT lives in: ONE consumer) receives the reassembled outputs.
        Every rank pushes its shard into rank 0's buffer alone: 1 / world of the all-gather's bytes per link, and rank 0's seven
        links carry one shard each in parallel.  Rank 0 waits for all `world` flags as before and then stores the step number into
        every rank's back-flag slot; a rank does not push step i + 1 before it has seen rank 0's back-flag for step i (raised in rank
        0's stream behind its consumption of step i - 1, whose buffer parity step i + 1 reuses), so no rank runs ahead of the consumer
        by more than one step.  The other ranks get (None, ...)."""
        import ctypes as C
        if cb.device.type != "cuda":
            raise RuntimeError("PeerExchange needs the HIP engine (peer stores between device buffers)")
        self.cb = cb
        self.engine = engine
        self.root_only = bool(root_only)
        if engine is not None and engine.kernels_per_eval() != 1:
            raise RuntimeError("PeerExchange(engine=...): the shard's cost must be summed inside the knot launch (shards of at most 256 knots)")
        self.world, self.rank = cb.world, cb.rank
        lib = cb._lib
        self._lib = lib
        vp, i64 = C.c_void_p, C.c_int64
        lib.hipnlp_ipc_alloc.argtypes = [C.c_size_t, C.c_int, C.POINTER(vp), C.c_char_p]
        lib.hipnlp_ipc_open.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp)]
        lib.hipnlp_ipc_close.argtypes = [vp]
        lib.hipnlp_ipc_free.argtypes = [vp]
        lib.hipnlp_peer_push.argtypes = [vp, vp, i64, vp, C.c_int, vp]
        lib.hipnlp_peer_signal.argtypes = [vp, C.c_int, C.c_int, C.c_ulonglong, vp]
        lib.hipnlp_peer_signal_checked.argtypes = [vp, C.c_int, C.c_int, C.c_ulonglong, vp, vp]
        lib.hipnlp_peer_wait.argtypes = [vp, C.c_int, C.c_ulonglong, vp, i64, vp, vp]
        self.tot = cb.n + cb.nnz + cb.m
        self.olen = self.tot + self.world + 1             # [grad | jac | g | f partials (world) | f]
        dev_index = cb.device.index if cb.device.index is not None else torch.cuda.current_device()
        self.dev_index = dev_index
        # this rank's buffers: two output buffers and one flag array, in ONE allocation (one IPC handle)
        self.flag_words = 64
        nbytes = 8 * (2 * self.olen + self.flag_words)
        # Every failure below is made COLLECTIVE before anybody raises: a rank that threw on its own would leave the others waiting in
        # the next collective for ever.
        mine = vp()
        handle = C.create_string_buffer(64)
        mine_ok = lib.hipnlp_ipc_alloc(C.c_size_t(nbytes), dev_index, C.byref(mine), handle) == 0
        self._mine = mine.value if mine_ok else None
        handles = [None] * self.world
        if self.world > 1:
            dist.all_gather_object(handles, bytes(handle.raw) if mine_ok else None, group=cb.group)
        else:
            handles = [bytes(handle.raw) if mine_ok else None]
        self._opened = []
        bases = []
        opened_ok = all(h is not None for h in handles)
        if opened_ok:
            for r in range(self.world):
                if r == self.rank:
                    bases.append(self._mine)
                    continue
                p = vp()
                if lib.hipnlp_ipc_open(handles[r], dev_index, C.byref(p)) != 0:
                    opened_ok = False
                    break
                self._opened.append(p.value)
                bases.append(p.value)
        oks = [opened_ok]
        if self.world > 1:
            oks = [None] * self.world
            dist.all_gather_object(oks, opened_ok, group=cb.group)
        if not all(oks):
            for p in self._opened:
                lib.hipnlp_ipc_close(p)
            if self._mine:
                lib.hipnlp_ipc_free(self._mine)
            self._mine = None
            raise RuntimeError("peer exchange set-up failed on rank(s) %s (hipnlp_ipc_alloc / hipnlp_ipc_open: no peer access between the devices?)"
                               % [r for r, ok in enumerate(oks) if not ok])
        dev = cb.device
        # device arrays of pointers: output buffer of every rank per parity, flag array of every rank
        self.peer_out = [torch.tensor([b + 8 * par * self.olen for b in bases], dtype=torch.int64, device=dev) for par in (0, 1)]
        self.peer_flags = torch.tensor([b + 8 * 2 * self.olen for b in bases], dtype=torch.int64, device=dev)
        self.my_flags = self._mine + 8 * 2 * self.olen
        # destination of every entry of MY fused shard buffer in the output buffer (the inverse of the reassembly index)
        src = cb.src.cpu().numpy()
        dst = np.full(cb.shard_len, -1, dtype=np.int64)
        lo, hi = self.rank * cb.shard_len, (self.rank + 1) * cb.shard_len
        own = np.nonzero((src >= lo) & (src < hi))[0]
        dst[src[own] - lo] = own
        dst[0] = self.tot + self.rank                    # the cost partial
        self.dst = torch.from_numpy(dst).to(dev)
        self.status = torch.zeros(1, dtype=torch.int32, device=dev)   # sticky: raised by a wait that gave up, cleared only here
        self.seq = 0
        self._views = []
        for par in (0, 1):   # torch views of this rank's own output buffers
            self._views.append(_device_view(self._mine + 8 * par * self.olen, self.olen, dev))
        # flag words of a rank: [0, world) step flags | [world, 2 world) back-flags (gather_to_root: slot 0, written by rank 0) |
        # [32, 32 + world) handshake
        if 2 * self.world > 32 or 32 + self.world > self.flag_words:
            raise RuntimeError("peer exchange: at most 16 ranks")
        self.root_out = [torch.tensor([bases[0] + 8 * par * self.olen], dtype=torch.int64, device=dev) for par in (0, 1)]
        self.root_flags = torch.tensor([bases[0] + 8 * 2 * self.olen], dtype=torch.int64, device=dev)
        self.back_flags = torch.tensor([b + 8 * (2 * self.olen + self.world) for b in bases], dtype=torch.int64, device=dev)
        self.my_back_flags = self._mine + 8 * (2 * self.olen + self.world)
        self._scratch = torch.zeros(4, dtype=torch.float64, device=dev)
        if self.world > 1:
            dist.barrier(group=cb.group)                  # every rank has opened every buffer
        self._handshake(bases, handshake_timeout_s)
        self.refresh_constants()

    def refresh_constants(self):
        """(compact exchange: the callback was built with const_mask / const_fill) the constant entries of jac g into this rank's OWN two
        output buffers — the peers store the varying runs around them.  At set-up and after every change of the parameters; a rank
        that receives nothing (gather_to_root, rank != 0) holds nothing.  Collective-free: every rank fills its own memory."""
        cb = self.cb
        self._filled_gen = cb.param_gen() if cb.param_gen is not None else None
        if not cb.compact or (self.root_only and self.rank != 0):
            return
        with torch.cuda.stream(cb.stream):
            for par in (0, 1):
                cb.const_fill(self._views[par][cb.n:cb.n + cb.nnz], cb.stream.cuda_stream)
        cb.stream.synchronize()

    def _handshake(self, bases, timeout_s):
        """One word written into every peer's buffer and one word read from every peer, BEFORE anything is timed or trusted: mapping
        a peer's allocation (hipnlp_ipc_open) says nothing about stores arriving there.  Every rank stores a token into its slot of
        every rank's handshake words (the store path of the exchange itself: hipnlp_peer_signal) and waits — bounded by the wait
        kernel, and by `timeout_s` on the host — for all `world` tokens in its own; the outcome is agreed on collectively, so a link
        that does not deliver fails loudly on EVERY rank instead of hanging one of them."""
        import time
        cb, lib, dev = self.cb, self._lib, self.cb.device
        token = 0x48414E44 + self.world   # the same on every rank
        hs = torch.tensor([b + 8 * (2 * self.olen + 32) for b in bases], dtype=torch.int64, device=dev)
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        with torch.cuda.stream(cb.stream):
            sh = cb.stream.cuda_stream
            rc = lib.hipnlp_peer_signal(hs.data_ptr(), self.world, self.rank, token, sh)
            rc |= lib.hipnlp_peer_wait(self._mine + 8 * (2 * self.olen + 32), self.world, token, self._scratch.data_ptr(), 0, status.data_ptr(), sh)
            done = torch.cuda.Event()
            done.record(cb.stream)
        t0 = time.time()
        while not done.query() and time.time() - t0 < timeout_s:
            time.sleep(0.001)
        ok = rc == 0 and done.query() and int(status.item()) == 0
        oks = [ok]
        if self.world > 1:
            oks = [None] * self.world
            dist.all_gather_object(oks, ok, group=cb.group)
        if not all(oks):
            raise RuntimeError("peer exchange handshake failed on rank(s) %s: a word stored into a peer's buffer did not arrive "
                               "(peer access mapped but not working between these devices)" % [r for r, o in enumerate(oks) if not o])

    def __call__(self, x):
        cb = self.cb
        if cb.compact and cb.param_gen is not None and cb.param_gen() != getattr(self, "_filled_gen", None):
            self.refresh_constants()     # (set_params has run since this exchange's own two buffers were filled)
        cur = torch.cuda.current_stream(cb.device)
        if cur.cuda_stream != cb.stream.cuda_stream:
            cb.stream.wait_stream(cur)
        with torch.cuda.stream(cb.stream):
            sh = cb.stream.cuda_stream
            self.seq += 1
            par = self.seq & 1
            lib = self._lib
            rc = 0
            targets = self.root_out[par] if self.root_only else self.peer_out[par]
            ntargets = 1 if self.root_only else self.world
            if self.root_only and self.rank != 0 and self.seq > 1:
                # The buffer of this parity was last used by step seq - 2.  Rank 0 raises its back-flag for a step right behind its wait
                # for that step's flags — BEFORE its consumer has read the outputs — so the flag of step seq - 2 does not say the buffer
                # is free; the flag of step seq - 1 does: rank 0 raises it in its call for step seq - 1, which in rank 0's stream order
                # lies behind its consumption of step seq - 2.  (No cycle: rank 0's wait for step seq - 1 needs this rank's signal of
                # step seq - 1, sent in the previous call.)  A non-root rank is therefore never more than one step ahead of the consumer.
                rc |= lib.hipnlp_peer_wait(self.my_back_flags, 1, self.seq - 1, self._scratch.data_ptr(), 0, self.status.data_ptr(), sh)
            if self.engine is not None:
                # (compact exchange: the varying runs only, at their places in the pattern — the receivers' buffers hold the constants)
                (self.engine.eval_device_peers_vary if cb.compact else self.engine.eval_device_peers)(x.data_ptr(), targets.data_ptr(), ntargets, self.rank, stream=sh)
            else:
                cb.compute_shard(x, *cb.views, sh)
                rc |= lib.hipnlp_peer_push(cb.buf.data_ptr(), self.dst.data_ptr(), cb.shard_len, targets.data_ptr(), ntargets, sh)
            flags = self.root_flags if self.root_only else self.peer_flags
            # (checked: a rank whose own back-flag wait gave up has pushed into a buffer that may still have been read — it says so)
            rc |= lib.hipnlp_peer_signal_checked(flags.data_ptr(), ntargets, self.rank, self.seq, self.status.data_ptr(), sh)
            out = self._views[par]
            if not self.root_only or self.rank == 0:
                rc |= lib.hipnlp_peer_wait(self.my_flags, self.world, self.seq, out.data_ptr(), self.tot, self.status.data_ptr(), sh)
            if self.root_only and self.rank == 0:   # "step seq is complete here": slot 0 of every rank's back-flags
                rc |= lib.hipnlp_peer_signal(self.back_flags.data_ptr(), self.world, 0, self.seq, sh)
            if rc != 0:
                raise RuntimeError("peer exchange: a launch failed")
        if cur.cuda_stream != cb.stream.cuda_stream:
            cur.wait_stream(cb.stream)
        if self.root_only and self.rank != 0:
            return None, None, None, None
        n, nnz = cb.n, cb.nnz
        return out[self.tot + self.world], out[:n], out[n:n + nnz], out[n + nnz:self.tot]

    def bytes_sent_per_step(self):
        """bytes this rank stores into OTHER ranks' buffers per step (what its xGMI links carry)"""
        me = self.cb.infos[self.rank]
        rows = int((np.asarray(me["stage_rows"]).reshape(-1) >= 0).sum())
        shard = 8 * (me["glen"] + me["jvary" if self.cb.compact else "jlen"] + rows + 1)
        if self.root_only:
            return 0 if self.rank == 0 else shard
        return shard * (self.world - 1)

    def max_bytes_sent_per_step(self):
        """the largest of bytes_sent_per_step() over the ranks (what the busiest sender's links carry; rank 0 of a gather_to_root sends
        nothing, which says nothing about the exchange)"""
        best = 0
        for inf in self.cb.infos:
            rows = int((np.asarray(inf["stage_rows"]).reshape(-1) >= 0).sum())
            best = max(best, 8 * (inf["glen"] + inf["jvary" if self.cb.compact else "jlen"] + rows + 1))
        if self.root_only:
            return best if self.world > 1 else 0
        return best * (self.world - 1)

    def timed_out(self):
        """True if ANY wait since set-up gave up (a rank never signalled; sticky: the flag is only raised on the device): the outputs
        of such a step are NaN throughout.  Synchronises the callback's stream."""
        self.cb.stream.synchronize()
        return bool(self.status.item())

    def check(self):
        """raises if a wait has given up since set-up"""
        if self.timed_out():
            raise RuntimeError("peer exchange: a rank did not signal a step in time; the outputs of that step were poisoned (NaN)")

    def close(self, barrier=True):
        """barrier=False: the caller has just been through a barrier of its own behind the last step (every rank's pushes are complete)"""
        if getattr(self, "_mine", None):
            self.cb.stream.synchronize()
            if self.world > 1 and barrier:
                dist.barrier(group=self.cb.group)         # nobody still pushes into a buffer about to go
            for p in self._opened:
                self._lib.hipnlp_ipc_close(p)
            self._views = []
            self._lib.hipnlp_ipc_free(self._mine)
            self._mine = None


def batch_range(batch, world, rank):
    """Trajectories [begin, end) of rank `rank` when `batch` independent trajectories (initial guesses of one receding-horizon step,
    BASELINE config 5) are dealt over `world` ranks: contiguous, equal shares (the collectives below move equal-sized pieces)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    if batch % world != 0:
        raise ValueError(f"{batch} trajectories cannot be dealt evenly over {world} ranks")
    q = batch // world
    return rank * q, (rank + 1) * q


class BatchDealtCallback:
    """BASELINE config 5's multi-GPU form (SURVEY §8e "batch x knot flattened, then sharded", with the cut between trajectories):
    `batch` independent trajectories dealt over the ranks (batch_range), every rank evaluates its own [b0, b1) with ONE launch of a
    batched handle, and the outputs of ALL trajectories end up on rank 0 — the process the NLP drivers of the guesses live in.
    No entry needs re-ordering: the fused buffer of a rank is [f (Bl) | grad (Bl n) | g (Bl m) | jac (Bl nnz)] and every trajectory's
    four outputs are contiguous pieces of it, so rank 0 hands out VIEWS of the gathered buffers (`trajectory(b)`).

        compute_batch(x_local, f_view, grad_view, g_view, jac_view, stream_handle)     fills the rank's views
        to_root(x_local)   ONE collective (dist.gather: RCCL over xGMI, gloo in the CPU tests); returns self on rank 0, None elsewhere

    Without the constants of jac g (`const_mask` + `const_fill`, as ShardedCallback): the jac view of a rank is [Bl][nvary] — the varying
    runs of every trajectory's knot blocks — and so is what the gather moves; rank 0 keeps a COMPLETE [batch][nnz] value array whose
    constant entries arrive ONCE per parameter set (`refresh_constants()`: a one-off gather of every rank's constants — trajectories may
    differ in dt and mass) and scatters the varying entries into it behind every gather (one launch): `trajectory(b)` hands out views of
    it as before."""

    def __init__(self, batch, n, m, nnz, compute_batch, device, group=None, const_mask=None, const_fill=None, param_gen=None):
        """param_gen() -> generation of the engine's parameters: refresh_constants() here is a COLLECTIVE (every rank's constants travel
        to rank 0), so a stale buffer cannot be repaired by the rank that notices — to_root raises instead of returning a Jacobian with
        the constants of the previous parameters."""
        self.param_gen = param_gen
        self._filled_gen = None
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.batch, self.n, self.m, self.nnz = batch, n, m, nnz
        self.b0, self.b1 = batch_range(batch, self.world, self.rank)
        self.local = self.b1 - self.b0
        self.compute_batch = compute_batch
        self.device = torch.device(device)
        Bl = self.local
        self.compact = const_mask is not None
        self.const_fill = const_fill
        self.jw = nnz                                    # entries of a trajectory's jac in the exchange buffer
        if self.compact:
            if const_fill is None:
                raise ValueError("const_mask needs const_fill")
            const_mask = np.asarray(const_mask, dtype=bool)
            if const_mask.shape != (nnz,):
                raise ValueError("const_mask must have one flag per entry of the pattern")
            self.jw = int((~const_mask).sum())
            self.vary_idx = torch.from_numpy(np.nonzero(~const_mask)[0]).to(device)
        self.o_grad, self.o_g, self.o_jac = Bl, Bl + Bl * n, Bl + Bl * (n + m)
        self.shard_len = Bl * (1 + n + m + self.jw)
        self.full_len = Bl * (1 + n + m + nnz)           # a rank's piece of a buffer that holds complete Jacobians (BatchPeerToRoot)
        self.buf = torch.zeros(self.shard_len, dtype=torch.float64, device=device)
        self.all = torch.zeros(self.world * self.shard_len, dtype=torch.float64, device=device) if self.rank == 0 else None
        self.full_jac = torch.zeros(batch, nnz, dtype=torch.float64, device=device) if (self.compact and self.rank == 0) else None
        self.views = self._views_of(self.buf)
        self.stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None
        self.refresh_constants()

    def _views_of(self, buf, complete=False):
        Bl, n, m = self.local, self.n, self.m
        jw = self.nnz if complete else self.jw
        return (buf[0:Bl], buf[self.o_grad:self.o_g].view(Bl, n), buf[self.o_g:self.o_jac].view(Bl, m), buf[self.o_jac:self.o_jac + Bl * jw].view(Bl, jw))

    def refresh_constants(self):
        """(compact exchange) every rank's constants — of ITS trajectories, under the parameters last set — into rank 0's complete
        array: one gather per parameter set (a collective: every rank calls it, at set-up and after every set_params)"""
        self._filled_gen = self.param_gen() if self.param_gen is not None else None
        if not self.compact:
            return
        mine = torch.zeros(self.local, self.nnz, dtype=torch.float64, device=self.device)
        if self.stream is None:
            self.const_fill(mine, 0)
        else:
            cur = torch.cuda.current_stream(self.device)
            self.stream.wait_stream(cur)
            with torch.cuda.stream(self.stream):
                self.const_fill(mine, self.stream.cuda_stream)
            self.stream.synchronize()
        if self.world == 1:
            self.full_jac.copy_(mine)
            return
        on_host = self.device.type == "cuda" and dist.get_backend(self.group) == "gloo"   # (rehearsal: gloo gathers host tensors)
        piece = mine.cpu() if on_host else mine
        if self.rank == 0:
            parts = [torch.empty_like(piece) for _ in range(self.world)]
            dist.gather(piece, parts, dst=0, group=self.group)
            self.full_jac.copy_(torch.cat(parts).to(self.device))
        else:
            dist.gather(piece, None, dst=0, group=self.group)

    def bytes_sent_per_step(self):
        return 0 if self.rank == 0 else 8 * self.shard_len

    def max_bytes_sent_per_step(self):
        return 8 * self.shard_len if self.world > 1 else 0

    def trajectory(self, b, gathered=None):
        """(f, grad, g, jac) views of trajectory b on rank 0: of the last to_root's result, or — `gathered` — of a buffer that holds the
        ranks' pieces with COMPLETE Jacobians (BatchPeerToRoot)"""
        r, i = divmod(b, self.local)
        if gathered is not None:
            f, grad, g, jac = self._views_of(gathered[r * self.full_len:(r + 1) * self.full_len], complete=True)
            return f[i], grad[i], g[i], jac[i]
        f, grad, g, jac = self._views_of(self.all[r * self.shard_len:(r + 1) * self.shard_len])
        return f[i], grad[i], g[i], (self.full_jac[b] if self.compact else jac[i])

    def _expand(self):
        """(rank 0, compact exchange) the gathered varying entries into their places of the complete array: one launch"""
        if self.compact:
            got = self.all.view(self.world, self.shard_len)[:, self.o_jac:].reshape(self.world, self.local, self.jw)
            self.full_jac.view(self.world, self.local, self.nnz).index_copy_(2, self.vary_idx, got)

    def _run(self, x_local, stream_handle):
        self.compute_batch(x_local, *self.views, stream_handle)
        if self.world == 1:
            self.all.copy_(self.buf)
            self._expand()
            return self
        chunks = list(self.all.view(self.world, self.shard_len).unbind(0)) if self.rank == 0 else None
        if self.device.type == "cuda" and dist.get_backend(self.group) == "gloo":   # rehearsal: several ranks on one GPU (gloo gathers host tensors)
            torch.cuda.current_stream(self.device).synchronize()
            host = [torch.empty(self.shard_len, dtype=torch.float64) for _ in range(self.world)] if self.rank == 0 else None
            dist.gather(self.buf.cpu(), host, dst=0, group=self.group)
            if self.rank == 0:
                self.all.copy_(torch.cat(host))
        else:
            dist.gather(self.buf, chunks, dst=0, group=self.group)
        if self.rank == 0:
            self._expand()
        return self if self.rank == 0 else None

    def _check_fresh(self):
        if self.compact and self.param_gen is not None and self.param_gen() != self._filled_gen:
            raise RuntimeError("the parameters changed (set_params) since the constants of jac g were gathered: call refresh_constants() "
                               "on EVERY rank (a collective) before the next exchange")

    def to_root(self, x_local):
        self._check_fresh()
        if self.stream is None:
            return self._run(x_local, 0)
        cur = torch.cuda.current_stream(self.device)
        if cur.cuda_stream == self.stream.cuda_stream:
            return self._run(x_local, self.stream.cuda_stream)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            out = self._run(x_local, self.stream.cuda_stream)
        cur.wait_stream(self.stream)
        return out

    def local_only(self, x_local):
        """the rank's trajectories evaluated into its own buffer, nothing exchanged (what the gather costs on top)"""
        if self.stream is None:
            self.compute_batch(x_local, *self.views, 0)
            return
        cur = torch.cuda.current_stream(self.device)
        if cur.cuda_stream != self.stream.cuda_stream:
            self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            self.compute_batch(x_local, *self.views, self.stream.cuda_stream)
        if cur.cuda_stream != self.stream.cuda_stream:
            cur.wait_stream(self.stream)


class BatchPeerToRoot:
    """BatchDealtCallback's gather without a collective: rank 0 owns ONE device buffer [parity][rank][f | grad | g | jac] shared through a
    HIP IPC handle, and every rank's batched knot kernel stores its trajectories' outputs straight into its piece of it over its own
    xGMI link (hipnlp_eval_device with device-visible addresses of rank 0's buffer: the kernel needs no change, a trajectory's outputs
    are contiguous runs).  Flags as in PeerExchange(root_only): a rank signals a step, rank 0 waits for all `world` signals and raises
    the back-flag a rank waits for before it stores the step after the next one.  Returns the BatchDealtCallback (views of the step's
    parity through `trajectory(b, gathered)`) on rank 0, None elsewhere."""

    def __init__(self, bc, engine, handshake_timeout_s=5.0):
        import ctypes as C
        from .hipnlp import load_library
        if bc.device.type != "cuda":
            raise RuntimeError("BatchPeerToRoot needs the HIP engine")
        self.bc, self.engine = bc, engine
        self.world, self.rank = bc.world, bc.rank
        lib = load_library()
        self._lib = lib
        vp, i64 = C.c_void_p, C.c_int64
        lib.hipnlp_ipc_alloc.argtypes = [C.c_size_t, C.c_int, C.POINTER(vp), C.c_char_p]
        lib.hipnlp_ipc_open.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp)]
        lib.hipnlp_ipc_close.argtypes = [vp]
        lib.hipnlp_ipc_free.argtypes = [vp]
        lib.hipnlp_peer_signal.argtypes = [vp, C.c_int, C.c_int, C.c_ulonglong, vp]
        lib.hipnlp_peer_signal_checked.argtypes = [vp, C.c_int, C.c_int, C.c_ulonglong, vp, vp]
        lib.hipnlp_peer_wait.argtypes = [vp, C.c_int, C.c_ulonglong, vp, i64, vp, vp]
        if 2 * self.world > 32:
            raise RuntimeError("peer exchange: at most 16 ranks")
        self.dev_index = bc.device.index if bc.device.index is not None else torch.cuda.current_device()
        # every rank allocates (its flag words are written by rank 0); only rank 0's allocation holds output buffers
        self.olen = self.world * bc.full_len + self.world + 1        # [rank][f | grad | g | jac (complete)] | scratch of the wait kernel
        self.flag_words = 64
        words = (2 * self.olen if self.rank == 0 else 0) + self.flag_words
        mine, handle = vp(), C.create_string_buffer(64)
        mine_ok = lib.hipnlp_ipc_alloc(C.c_size_t(8 * words), self.dev_index, C.byref(mine), handle) == 0
        self._mine = mine.value if mine_ok else None
        handles = [bytes(handle.raw) if mine_ok else None]
        if self.world > 1:
            handles = [None] * self.world
            dist.all_gather_object(handles, bytes(handle.raw) if mine_ok else None, group=bc.group)
        self._opened, bases = [], []
        opened_ok = all(h is not None for h in handles)
        if opened_ok:
            for r in range(self.world):
                if r == self.rank:
                    bases.append(self._mine)
                    continue
                if self.rank != 0 and r != 0:
                    bases.append(0)            # (only rank 0 talks to everybody; the others talk to rank 0)
                    continue
                p = vp()
                if lib.hipnlp_ipc_open(handles[r], self.dev_index, C.byref(p)) != 0:
                    opened_ok = False
                    break
                self._opened.append(p.value)
                bases.append(p.value)
        oks = [opened_ok]
        if self.world > 1:
            oks = [None] * self.world
            dist.all_gather_object(oks, opened_ok, group=bc.group)
        if not all(oks):
            self._release()
            raise RuntimeError("batch peer exchange set-up failed on rank(s) %s (hipnlp_ipc_alloc / hipnlp_ipc_open)" % [r for r, ok in enumerate(oks) if not ok])
        dev = bc.device
        flag_off = [8 * (2 * self.olen if r == 0 else 0) for r in range(self.world)]
        root = bases[0]
        self.root_flags = torch.tensor([root + flag_off[0]], dtype=torch.int64, device=dev)            # step flags live on rank 0: [0, world)
        self.my_flags = self._mine + flag_off[self.rank]
        self.my_back_flags = self.my_flags + 8 * self.world                                           # back-flag slot 0 of every rank
        self.back_flags = torch.tensor([b + fo + 8 * self.world for b, fo in zip(bases, flag_off)], dtype=torch.int64, device=dev) if self.rank == 0 else None
        self.root_out = [root + 8 * par * self.olen for par in (0, 1)]
        self.status = torch.zeros(1, dtype=torch.int32, device=dev)
        self._scratch = torch.zeros(4, dtype=torch.float64, device=dev)
        self.seq = 0
        self._gathered = [_device_view(self._mine + 8 * par * self.olen, self.olen, dev) for par in (0, 1)] if self.rank == 0 else None
        if self.world > 1:
            dist.barrier(group=bc.group)
        self._handshake(bases, flag_off, handshake_timeout_s)

    def _handshake(self, bases, flag_off, timeout_s):
        """a token through the store path of the exchange, both ways, before anything is timed or trusted: every rank stores into its
        slot of rank 0's handshake words, rank 0 into slot 0 of everybody's; the outcome is agreed on collectively"""
        import time
        bc, lib, dev = self.bc, self._lib, self.bc.device
        token = 0x42415443 + self.world
        with torch.cuda.stream(bc.stream):
            sh = bc.stream.cuda_stream
            to_root = torch.tensor([bases[0] + flag_off[0] + 8 * 32], dtype=torch.int64, device=dev)
            rc = lib.hipnlp_peer_signal(to_root.data_ptr(), 1, self.rank, token, sh)
            status = torch.zeros(1, dtype=torch.int32, device=dev)
            if self.rank == 0:
                rc |= lib.hipnlp_peer_wait(self.my_flags + 8 * 32, self.world, token, self._scratch.data_ptr(), 0, status.data_ptr(), sh)
                back = torch.tensor([b + fo + 8 * 48 for b, fo in zip(bases, flag_off)], dtype=torch.int64, device=dev)
                rc |= lib.hipnlp_peer_signal(back.data_ptr(), self.world, 0, token, sh)
            rc |= lib.hipnlp_peer_wait(self.my_flags + 8 * 48, 1, token, self._scratch.data_ptr(), 0, status.data_ptr(), sh)
            done = torch.cuda.Event()
            done.record(bc.stream)
        t0 = time.time()
        while not done.query() and time.time() - t0 < timeout_s:
            time.sleep(0.001)
        ok = rc == 0 and done.query() and int(status.item()) == 0
        oks = [ok]
        if self.world > 1:
            oks = [None] * self.world
            dist.all_gather_object(oks, ok, group=bc.group)
        if not all(oks):
            raise RuntimeError("batch peer exchange handshake failed on rank(s) %s" % [r for r, o in enumerate(oks) if not o])

    def __call__(self, x_local):
        bc, lib = self.bc, self._lib
        cur = torch.cuda.current_stream(bc.device)
        if cur.cuda_stream != bc.stream.cuda_stream:
            bc.stream.wait_stream(cur)
        with torch.cuda.stream(bc.stream):
            sh = bc.stream.cuda_stream
            self.seq += 1
            par = self.seq & 1
            rc = 0
            if self.rank != 0 and self.seq > 1:   # (as PeerExchange(root_only): rank 0's back-flag of step seq - 1 lies behind its consumption of step seq - 2)
                rc |= lib.hipnlp_peer_wait(self.my_back_flags, 1, self.seq - 1, self._scratch.data_ptr(), 0, self.status.data_ptr(), sh)
            # (a varying-first engine fills the constants of its piece at its first sight of it — once per parity and parameter set, over its
            #  link — and stores the varying runs from then on: hipnlp_eval_device)
            base = self.root_out[par] + 8 * self.rank * bc.full_len
            self.engine.eval_device(x_local.data_ptr(), base, base + 8 * bc.o_grad, base + 8 * bc.o_g, base + 8 * bc.o_jac, stream=sh)
            rc |= lib.hipnlp_peer_signal_checked(self.root_flags.data_ptr(), 1, self.rank, self.seq, self.status.data_ptr(), sh)
            if self.rank == 0:
                out = self._gathered[par]
                rc |= lib.hipnlp_peer_wait(self.my_flags, self.world, self.seq, out.data_ptr(), self.world * bc.full_len, self.status.data_ptr(), sh)
                rc |= lib.hipnlp_peer_signal(self.back_flags.data_ptr(), self.world, 0, self.seq, sh)
            if rc != 0:
                raise RuntimeError("batch peer exchange: a launch failed")
        if cur.cuda_stream != bc.stream.cuda_stream:
            cur.wait_stream(bc.stream)
        return self._gathered[par][:self.world * bc.full_len] if self.rank == 0 else None

    def _piece_bytes(self):
        """what a rank's launch stores into rank 0's buffer per step: complete Jacobians, or — an engine whose destinations hold the
        constants (varying-first handle, hipnlp_set_constant_jacobian on: the default) — their varying runs only"""
        bc = self.bc
        skips = getattr(self.engine, "jac_varying_first", False) and getattr(self.engine, "constants_in_place", True)
        jw = self.engine.jac_vary_layout()["total"] if skips else bc.nnz
        return 8 * bc.local * (1 + bc.n + bc.m + jw)

    def bytes_sent_per_step(self):
        return 0 if self.rank == 0 else self._piece_bytes()

    def max_bytes_sent_per_step(self):
        return self._piece_bytes() if self.world > 1 else 0

    def timed_out(self):
        self.bc.stream.synchronize()
        return bool(self.status.item())

    def _release(self):
        for p in getattr(self, "_opened", []):
            self._lib.hipnlp_ipc_close(p)
        self._opened = []
        if getattr(self, "_mine", None):
            self._lib.hipnlp_ipc_free(self._mine)
        self._mine = None

    def close(self, barrier=True):
        if getattr(self, "_mine", None):
            self.bc.stream.synchronize()
            if self.world > 1 and barrier:
                dist.barrier(group=self.bc.group)
            self._gathered = None
            self._release()


def hip_batch_backend(engine, compact=False):
    """compute_batch backed by a batched HIP engine handle (hipnlp_eval_device on the rank's trajectories).  compact: the jac view is
    [Bl][nvary] — the varying runs of every trajectory's knot blocks (hipnlp_eval_device_vary; a varying-first handle)"""
    def compute(x_local, f_view, grad_view, g_view, jac_view, stream_handle):
        if not stream_handle:
            raise RuntimeError("the batch-dealt path needs an explicit (non-default) stream")
        (engine.eval_device_vary if compact else engine.eval_device)(x_local.data_ptr(), f_view.data_ptr(), grad_view.data_ptr(), g_view.data_ptr(), jac_view.data_ptr(),
                                                                     stream=stream_handle)
    return compute


def _device_view(ptr, count, device):
    """float64 torch view of `count` doubles of device memory that torch did not allocate (the owner keeps it alive)"""
    class _Holder:
        pass
    h = _Holder()
    h.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 2}
    return torch.as_tensor(h, device=device)


def hip_shard_backend(engine, compact=False):
    """compute_shard backed by the HIP engine handle that owns this rank's knots.  compact: the jac view receives the varying runs of the
    rank's knot blocks only (hipnlp_eval_device_shard_vary; a handle created with jac_varying_first=True)."""
    def compute(x, f_view, grad_view, jac_view, stage_view, stream_handle):
        if not stream_handle:
            raise RuntimeError("the sharded path needs an explicit (non-default) stream: a NULL stream means different streams to "
                               "hipnlp_eval_device_shard and hipnlp_reassemble")
        (engine.eval_device_shard_vary if compact else engine.eval_device_shard)(x.data_ptr(), f_view.data_ptr(), grad_view.data_ptr(), stage_view.data_ptr(),
                                                                                  jac_view.data_ptr(), stream=stream_handle)
    return compute


def hip_shard_info(engine, knot_begin, knot_end, compact=False):
    """shard_info of ShardedCallback for a HIP engine handle.  compact (a varying-first handle): the record of a compact exchange — what
    hipnlp_eval_device_shard_vary fills: the varying runs of jac g ("jvary") and a staging of g that lists the rows the rank's knots own
    behind one another ("slen" rows, all of stage_rows >= 0) instead of G_STAGE native slots per knot"""
    d = engine.dims
    rows = np.stack([engine.stage_rows(k) for k in range(knot_begin, knot_end)])
    info = {"glen": int(d.shard_grad), "jlen": int(d.shard_nnz), "nk": knot_end - knot_begin, "stage_rows": rows}
    if compact:
        info["jvary"] = engine.jac_vary_layout()["shard_len"]
        info["stage_rows"] = rows[rows >= 0]          # (row-major: knot behind knot, slot order inside a knot)
        info["slen"] = int(d.shard_g_rows)
        if info["stage_rows"].size != info["slen"]:
            raise RuntimeError("staging rows of the shard: %d, hipnlp_get_dims says %d" % (info["stage_rows"].size, info["slen"]))
    return info


def hip_constants(engine):
    """const_mask / const_fill of a varying-first HIP engine handle for ShardedCallback and BatchDealtCallback: the mask of the handle's
    pattern, and hipnlp_fill_jac_constants on the caller's view (whole horizon: a shard handle knows the whole pattern)"""
    def fill(jac_view, stream_handle):
        engine.fill_jac_constants(jac_view.data_ptr(), True, stream_handle)
    # (param_gen: the receiving buffers hold the constants of ONE parameter set; the callbacks compare it at every call)
    return {"const_mask": engine.jac_constant_mask(), "const_fill": fill, "param_gen": lambda: engine.params_generation}


class HostSink:
    """One shared, pinned host buffer [f_parts (world, padded to 8) | grad (n) | jac (nnz) | g (m)] that every rank's knot kernel
    stores into directly (no collective, no staging copy): SURVEY §5's "each rank D2H's its own slice into one pinned host buffer",
    what a CPU-side IPOPT process wants.  The segment lives in /dev/shm; every rank maps it and registers the mapping with the HIP
    runtime (hipHostRegister), which yields the device-visible address the kernel stores to.

    name: shared-memory name all ranks agree on (rank 0 creates, the others open after a barrier the caller provides through
    `barrier()`); world = 1 needs no barrier."""

    def __init__(self, name, n, m, nnz, world, rank, barrier=None, agree=None):
        """agree(ok) -> bool: True iff EVERY rank passed ok (a collective; replaces the second barrier).  Without it a rank whose
        registration fails raises alone and leaves the others in the barrier."""
        import ctypes as C
        from .hipnlp import load_library
        self.n, self.m, self.nnz, self.world, self.rank = n, m, nnz, world, rank
        self.o_grad = 8 * ((world + 7) // 8)
        self.o_jac = self.o_grad + n
        self.o_g = self.o_jac + nnz
        self.o_x = self.o_g + m                              # x of the step in flight, written by rank 0 (the one caller), read by every rank's kernel
        self.o_mail = 8 * ((self.o_x + n + 7) // 8)          # int64 words: [0] step number rank 0 has posted x for, [8 + r] step rank r has completed
        self.count = self.o_mail + 8 + 8 * ((world + 7) // 8)
        nbytes = ((self.count * 8 + 4095) // 4096) * 4096
        self.path = os.path.join("/dev/shm", name)
        self.dev = self.host = self._mm = None
        rc, failure = -1, None
        # Whatever goes wrong on this rank (mapping, loading the library, registering), it still takes part in the barrier / agreement
        # below and rank 0 still removes the name: nothing is left behind in /dev/shm and nobody waits for a rank that raised alone.
        try:
            if rank == 0:
                fd = os.open(self.path, os.O_CREAT | os.O_RDWR | os.O_TRUNC, 0o600)
                try:
                    os.ftruncate(fd, nbytes)
                except Exception:
                    os.close(fd)
                    raise
        except Exception as err:  # noqa: BLE001
            failure = err
        if barrier is not None:
            barrier()
        try:
            if failure is None:
                if rank != 0:
                    fd = os.open(self.path, os.O_RDWR)
                try:
                    self._mm = mmap.mmap(fd, nbytes)
                finally:
                    os.close(fd)
                self.host = np.frombuffer(self._mm, dtype=np.float64, count=self.count)
                self.mail = np.frombuffer(self._mm, dtype=np.int64, count=self.count)[self.o_mail:]
                self._lib = load_library()
                self._lib.hipnlp_host_register.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
                self._lib.hipnlp_host_unregister.argtypes = [C.c_void_p]
                dev = C.c_void_p()
                self._addr = self.host.ctypes.data
                rc = self._lib.hipnlp_host_register(C.c_void_p(self._addr), C.c_size_t(nbytes), C.byref(dev))
                self.dev = dev.value if rc == 0 else None
        except Exception as err:  # noqa: BLE001
            failure = err
        try:
            if agree is not None:
                all_ok = agree(rc == 0 and failure is None)
            else:
                all_ok = rc == 0 and failure is None
                if barrier is not None:
                    barrier()   # every rank holds its mapping (or has given up): the name can go
        finally:
            if rank == 0 and os.path.exists(self.path):
                os.unlink(self.path)
        if not all_ok:
            self.close()
            raise RuntimeError("host sink set-up failed on this or another rank (here: %s)" % (failure if failure is not None else "hipnlp_host_register -> %d" % rc))

    def pointers(self):
        """device-visible addresses (f partial of this rank, grad, g, jac) for HipNlp.eval_device on a shard handle"""
        return (self.dev + 8 * self.rank, self.dev + 8 * self.o_grad, self.dev + 8 * self.o_g, self.dev + 8 * self.o_jac)

    # ---- one caller, one process per GPU: x in, "done" back, through the shared segment (no collective on the path) ------------------
    # Rank 0 is the process the NLP driver lives in.  A step: rank 0 writes x into the segment and posts the step number; every rank
    # (rank 0 too) sees the number, launches its shard's kernel — which READS x out of the segment over the rank's own PCIe link and
    # stores its outputs into it — waits for its own stream and writes the step number into its slot; rank 0 has the callback when all
    # `world` slots carry the number.  The multi-process sibling of hipnlp_multi_create (one process, one launching thread per shard).
    def x_pointer(self):
        """device-visible address of the x area: what every rank passes to HipNlp.eval_device as x"""
        return self.dev + 8 * self.o_x

    def post_x(self, x, step):
        """rank 0: x of step `step` (1, 2, ...) into the segment, then the step number (x86: stores are seen in program order)"""
        self.host[self.o_x:self.o_x + self.n] = np.asarray(x, dtype=np.float64).reshape(-1)
        self.mail[0] = step

    def wait_x(self, step, timeout_s=20.0):
        """every rank: until rank 0 has posted x of step `step`"""
        mail = self.mail
        if mail[0] >= step:
            return
        import time
        t_end = time.perf_counter() + timeout_s
        while mail[0] < step:
            if time.perf_counter() > t_end:
                raise TimeoutError("rank %d: x of step %d was never posted" % (self.rank, step))

    def done(self, step):
        """this rank's stores of step `step` are complete (call behind the synchronise of its stream)"""
        self.mail[8 + self.rank] = step

    def wait_all(self, step, timeout_s=20.0):
        """rank 0: until every rank has completed step `step`"""
        slots = self.mail[8:8 + self.world]
        import time
        t_end = time.perf_counter() + timeout_s
        while int(slots.min()) < step:
            if time.perf_counter() > t_end:
                raise TimeoutError("ranks %s never completed step %d" % ([r for r in range(self.world) if slots[r] < step], step))

    def views(self):
        """(f_parts[world], grad[n], jac[nnz], g[m]) numpy views of the shared buffer.
        VALID ONLY behind a cross-rank fence: every rank's kernel stores its own shard, so a reader needs every rank's stream
        synchronised AND a barrier between the ranks (a rank's own synchronise covers its own shard's stores only) — `sync(barrier)`."""
        h = self.host
        return h[:self.world], h[self.o_grad:self.o_jac], h[self.o_jac:self.o_g], h[self.o_g:self.o_x]

    def sync(self, barrier, synchronize):
        """the cross-rank fence views() / f() need: this rank's stream synchronised (`synchronize()`), then every rank through `barrier()`"""
        synchronize()
        if barrier is not None:
            barrier()

    def f(self):
        """sum of the rank partials in rank order (same fence requirement as views())"""
        parts = self.host[:self.world]
        tot = 0.0
        for r in range(self.world):   # rank order, as hipnlp_reassemble
            tot += float(parts[r])
        return tot

    def close(self):
        if getattr(self, "dev", None):
            import ctypes as C
            self._lib.hipnlp_host_unregister(C.c_void_p(self._addr))
        self.dev = None
        self.host = None
        self.mail = None
        mm, self._mm = getattr(self, "_mm", None), None
        if mm is not None:   # (whether or not the registration succeeded)
            try:
                mm.close()
            except BufferError:   # a numpy view is still alive somewhere: the mapping goes with it
                pass

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass
