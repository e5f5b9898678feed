"""World-size-2 `gloo` test of the knot-sharded callback plumbing (hippopt_amd/sharded.py): shard offsets,
fused shard buffer, all-gather, reassembly of [grad | jac | g] in reference order.  The per-rank compute is
emulated on the CPU (host emulation of the knot program, sliced to the rank's shard) because there is no GPU
here; on the GPU box the same class is driven by HipNlp.eval_device_shard (bench.py --gpus N)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _emulated_shard(emu, layout_rows, kb, ke, horizon, x_np, p_np):
    """What HipNlp.eval_device_shard produces for knots [kb, ke): partial f, grad shard, jac shard, g staging."""
    f, grad, g, jac, ct = emu.eval(x_np, p_np)
    ir, jc = emu.sparsity()
    col_knot = np.minimum(jc // 189, horizon - 1)  # the 6 global columns ride with the last knot
    sel = (col_knot >= kb) & (col_knot < ke)
    glen = 189 * (ke - kb) + (6 if ke == horizon else 0)
    stage = np.zeros((ke - kb, 550))
    for kk in range(ke - kb):
        rows = layout_rows[kb + kk]
        valid = rows >= 0
        stage[kk, valid] = g[rows[valid]]
    return grad[189 * kb:189 * kb + glen], jac[sel], stage


def _worker(rank, world, port, horizon, result_q, compact=False):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hippopt_amd.kinodyn_settings import periodic_step_settings
    from hippopt_amd.robot_model import synthetic_ergocub
    from hippopt_amd.sharded import ShardedCallback, knot_range
    from hippopt_amd.synthetic import make_workload
    from hostemu_lib import HostEmu
    from stage_rows_ref import stage_rows_from_blocks

    model = synthetic_ergocub()
    st = periodic_step_settings(horizon, model)
    x, p = make_workload(st, model, 1, 31)
    # compact: the exchange without the constants of jac g — handles in the varying-first order of a knot block (HIPNLP_FLAG_JAC_VARYING_FIRST),
    # the shards hand over the varying runs, every rank's reassembled buffer holds the constants (filled once per parameter set)
    emu = HostEmu(st, model, jac_varying_first=compact)
    rows_all = stage_rows_from_blocks(emu, horizon)
    kb, ke = knot_range(horizon, world, rank)
    ir, jc = emu.sparsity()
    col_knot = np.minimum(jc // 189, horizon - 1)
    sel = (col_knot >= kb) & (col_knot < ke)
    info = {"glen": 189 * (ke - kb) + (6 if ke == horizon else 0), "jlen": int(sel.sum()),
            "nk": ke - kb, "stage_rows": np.stack([rows_all[k] for k in range(kb, ke)])}
    mask = emu.constant_mask()
    params = {"p": p[0].copy(), "gen": 1}       # ("gen": what HipNlp.params_generation is for the HIP engine — bumped by every set_params)
    fills = []
    kw = {}
    if compact:
        info["jvary"] = int((sel & ~mask).sum())
        info["stage_rows"] = info["stage_rows"][info["stage_rows"] >= 0]     # the staging of g lists the owned rows only
        info["slen"] = int(info["stage_rows"].size)

        def const_fill(view, stream_handle):      # what hipnlp_fill_jac_constants does on the GPU: the constants under the parameters last set
            const = emu.constant_fill(params["p"])
            view[torch.from_numpy(mask)] = torch.from_numpy(const[mask])
            fills.append(1)
        kw = dict(const_mask=mask, const_fill=const_fill, param_gen=lambda: params["gen"])

    def compute(xt, f_view, grad_view, jac_view, stage_view, stream_handle):
        gs, js, stg = _emulated_shard(emu, rows_all, kb, ke, horizon, xt.numpy(), params["p"])
        if compact:
            js = js[~mask[sel]]                   # the varying runs of the rank's knot blocks, behind one another
            stg = stg[np.stack([rows_all[k] for k in range(kb, ke)]) >= 0]
        f_view[0] = float(rank + 1)  # partial costs: checked as a sum below
        grad_view.copy_(torch.from_numpy(np.ascontiguousarray(gs)))
        jac_view.copy_(torch.from_numpy(np.ascontiguousarray(js)))
        stage_view.copy_(torch.from_numpy(stg.reshape(-1)))

    cb = ShardedCallback(horizon, emu.n, emu.m, emu.nnz, info, compute, torch.device("cpu"), **kw)
    if compact:
        # what the all-gather moves: a third of the Jacobian less (0.43 of its entries are constants)
        plain_len = 1 + max(189 * (ke_ - kb_) + (6 if ke_ == horizon else 0) for kb_, ke_ in (knot_range(horizon, world, r) for r in range(world))) \
            + max(int(((col_knot >= kb_) & (col_knot < ke_)).sum()) for kb_, ke_ in (knot_range(horizon, world, r) for r in range(world))) \
            + max(ke_ - kb_ for kb_, ke_ in (knot_range(horizon, world, r) for r in range(world))) * 550
        assert cb.shard_len < plain_len and cb.bytes_sent_per_step() == 8 * cb.shard_len * (world - 1)
        assert 0.35 < mask.mean() < 0.5 and len(fills) == 1
    # the INPUT half: x exists in rank 0's host memory only (the process IPOPT lives in, opti_solver.py:479) — the other rank starts
    # from garbage and gets x through the feed (a broadcast), then every rank evaluates its shard from what it received
    from hippopt_amd.sharded import XFeed
    feed = XFeed(emu.n, torch.device("cpu"))
    x_in = feed.feed(feed.stage(x[0]) if rank == 0 else None)
    fed_ok = np.array_equal(x_in.numpy(), x[0])
    f, grad, jac, g = cb(x_in)
    f_ref, grad_ref, g_ref, jac_ref, _ = emu.eval(x[0], p[0])
    ok = (fed_ok and np.array_equal(grad.numpy(), grad_ref) and np.array_equal(jac.numpy(), jac_ref) and np.array_equal(g.numpy(), g_ref)
          and float(f) == sum(range(1, world + 1)))
    if rank == 0:
        try:
            feed.stage(x[0][:-1])
            ok = False
        except ValueError:
            pass
    # gather_to_root: one consumer — rank 0 alone receives the reassembled outputs (dist.gather), the others hold nothing
    x2 = x[0] + 1e-3
    f2, grad2, jac2, g2 = cb.to_root(torch.from_numpy(x2))
    if rank == 0:
        f_ref2, grad_ref2, g_ref2, jac_ref2, _ = emu.eval(x2, p[0])
        ok = ok and (np.array_equal(grad2.numpy(), grad_ref2) and np.array_equal(jac2.numpy(), jac_ref2) and np.array_equal(g2.numpy(), g_ref2)
                     and float(f2) == sum(range(1, world + 1)) and not np.array_equal(g_ref2, g_ref))
    else:
        ok = ok and f2 is None and grad2 is None and jac2 is None and g2 is None
    if compact:
        # a set_params that changes dt: the constants (-dt/2 entries of the trapezoid defects) change with it — refreshed, then every entry
        # of the reassembled Jacobian equals the emulation's under the NEW parameters (and differs from the old one at constant positions)
        p_new = p[0].copy()
        p_new[24 * horizon + 3 + 105 + 105] *= 1.3          # ParamOffsets::dt (layout.h)
        params["p"] = p_new
        params["gen"] += 1                                  # (a set_params; NO refresh_constants(): the callback notices and refreshes by itself)
        f3, grad3, jac3, g3 = cb(torch.from_numpy(x2))
        _, grad_ref3, g_ref3, jac_ref3, _ = emu.eval(x2, p_new)
        _, _, _, jac_old, _ = emu.eval(x2, p[0])
        ok = ok and np.array_equal(jac3.numpy(), jac_ref3) and np.array_equal(g3.numpy(), g_ref3) and np.array_equal(grad3.numpy(), grad_ref3)
        ok = ok and (jac_ref3[mask] != jac_old[mask]).any() and len(fills) == 2
        f4, grad4, jac4, g4 = cb.to_root(torch.from_numpy(x[0]))
        if rank == 0:
            ok = ok and np.array_equal(jac4.numpy(), emu.eval(x[0], p_new)[3])
        else:
            ok = ok and jac4 is None
    result_q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("horizon,compact", [(7, False), (10, False), (7, True), (10, True)])
def test_sharded_reassembly_world2(horizon, compact):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, horizon, q, compact)) for r in range(world)]
    for pr in procs:
        pr.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    assert sorted(results) == [(0, True), (1, True)]


def _batch_worker(rank, world, port, batch, horizon, result_q, compact=False):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hippopt_amd.kinodyn_settings import stairs_settings
    from hippopt_amd.robot_model import synthetic_ergocub
    from hippopt_amd.sharded import BatchDealtCallback, batch_range
    from hippopt_amd.synthetic import make_workload, place_on_step_flanks
    from hostemu_lib import HostEmu

    model = synthetic_ergocub()
    st = stairs_settings(horizon, model)
    x, p = make_workload(st, model, batch, 77)          # the same on every rank
    place_on_step_flanks(x, st, seed=77)
    emu = HostEmu(st, model, jac_varying_first=compact)
    b0, b1 = batch_range(batch, world, rank)
    mask = emu.constant_mask()
    if compact:
        p = p.copy()
        p[:, 24 * horizon + 3 + 105 + 105] *= 1.0 + 0.1 * np.arange(batch)     # every trajectory its own dt: its own constants

    def compute(xl, f_view, grad_view, g_view, jac_view, stream_handle):
        for i in range(b1 - b0):
            f, grad, g, jac, _ = emu.eval(xl[i].numpy(), p[b0 + i])
            f_view[i] = float(f)
            grad_view[i].copy_(torch.from_numpy(grad))
            g_view[i].copy_(torch.from_numpy(g))
            jac_view[i].copy_(torch.from_numpy(jac[~mask] if compact else jac))

    def const_fill(view, stream_handle):          # [local][nnz]: the constants of THIS rank's trajectories
        for i in range(b1 - b0):
            const = emu.constant_fill(p[b0 + i])
            view[i][torch.from_numpy(mask)] = torch.from_numpy(const[mask])
    gen = {"v": 1}
    kw = dict(const_mask=mask, const_fill=const_fill, param_gen=lambda: gen["v"]) if compact else {}
    bc = BatchDealtCallback(batch, emu.n, emu.m, emu.nnz, compute, torch.device("cpu"), **kw)
    ok = (bc.b0, bc.b1) == (b0, b1)
    # the guesses exist in rank 0's host memory (where their NLP drivers live): dealt to the ranks by a scatter
    from hippopt_amd.sharded import XDeal
    dealer = XDeal(batch, emu.n, torch.device("cpu"))
    for shift in (0.0, 1e-3):     # two steps: the second overwrites the first
        xl = dealer.deal(dealer.stage(x + shift) if rank == 0 else None)
        ok = ok and np.array_equal(xl.numpy(), x[b0:b1] + shift)
        got = bc.to_root(xl)
        if rank == 0:
            ok = ok and got is bc
            for b in range(batch):
                f, grad, g, jac = bc.trajectory(b)
                fr, gradr, gr, jacr, _ = emu.eval(x[b] + shift, p[b])
                ok = ok and float(f) == float(fr) and np.array_equal(grad.numpy(), gradr) and np.array_equal(g.numpy(), gr) and np.array_equal(jac.numpy(), jacr)
        else:
            ok = ok and got is None
    if compact:
        # a set_params without the (collective) refresh: the exchange refuses to hand out a Jacobian with the old constants, on every rank;
        # after the refresh it runs again
        gen["v"] += 1
        try:
            bc.to_root(torch.from_numpy(x[b0:b1]))
            ok = False
        except RuntimeError as err:
            ok = ok and "refresh_constants" in str(err)
        bc.refresh_constants()
        got = bc.to_root(torch.from_numpy(x[b0:b1]))
        ok = ok and ((got is bc) if rank == 0 else (got is None))
    jw = int((~mask).sum()) if compact else emu.nnz
    ok = ok and bc.max_bytes_sent_per_step() == 8 * (b1 - b0) * (1 + emu.n + emu.m + jw) and bc.bytes_sent_per_step() == (0 if rank == 0 else bc.max_bytes_sent_per_step())
    ok = ok and (not compact or jw < 0.7 * emu.nnz)
    result_q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("compact", [False, True])
def test_batch_dealt_over_the_ranks_reaches_rank_zero_world2(compact):
    """BASELINE config 5's multi-GPU form in small: independent trajectories (batched initial guesses on the stairs) dealt over two
    ranks, one collective to rank 0, every trajectory's outputs views of the gathered buffer — no reassembly.  compact: the gather moves
    the varying entries of jac g only; rank 0's complete array holds every trajectory's constants (gathered once per parameter set)"""
    world, batch, horizon = 2, 4, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_batch_worker, args=(r, world, port, batch, horizon, q, compact)) for r in range(world)]
    for pr in procs:
        pr.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    assert sorted(results) == [(0, True), (1, True)]


def test_batch_range_deals_evenly_or_refuses():
    from hippopt_amd.sharded import batch_range
    for world in (1, 2, 4, 8, 16):
        cover = []
        for r in range(world):
            a, b = batch_range(16, world, r)
            assert b - a == 16 // world
            cover += list(range(a, b))
        assert cover == list(range(16))
    with pytest.raises(ValueError):
        batch_range(16, 3, 0)


def test_knot_range_tiles_the_horizon():
    from hippopt_amd.sharded import knot_range
    for horizon in (7, 9, 100, 101, 800):
        for world in (1, 2, 4, 7, 8):
            if world > horizon:
                continue
            cover, sizes = [], []
            for r in range(world):
                a, b = knot_range(horizon, world, r)
                assert b > a   # never empty (hipnlp_create reads (0, 0) as "whole horizon" and rejects an empty shard)
                cover += list(range(a, b))
                sizes.append(b - a)
            assert cover == list(range(horizon)) and max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        knot_range(3, 4, 0)   # more ranks than knots: refused on every rank, before any collective


def test_overlapping_shards_are_refused():
    """every reassembly slot must be written exactly once: a row two shards claim is an error, not a silent overwrite"""
    from hippopt_amd.sharded import G_STAGE, ShardedCallback
    rows = np.full((1, G_STAGE), -1, np.int32)
    rows[0, :4] = [0, 1, 2, 2]   # row 2 twice, row 3 never
    info = {"glen": 3, "jlen": 5, "nk": 1, "stage_rows": rows}
    with pytest.raises(ValueError, match="written twice 1"):
        ShardedCallback(1, 3, 4, 5, info, lambda *a: None, torch.device("cpu"))
