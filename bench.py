#!/usr/bin/env python3
"""bench.py — NLP callback throughput (knots/s) of the hipnlp engine on MI355X.

One "step" = one complete callback set {f, grad f, g, jac g} of the ergoCub-shaped kinodynamic
multiple-shooting NLP (BASELINE.json metric) on synthetic, HBM-resident inputs.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--horizon 100] [--batch 1] [--shard knots|batch]

N = 1 : workload "kinodynamic periodic walking, N = 100 knots" (BASELINE config 4 on one GPU).
N > 1 : launched by torch.distributed.run, one rank per GPU.  Default `--shard batch`: N independent 100-knot trajectories, one per
        GPU, no data-path collective (the MPC / batched-initial-guess shape of BASELINE config 5: independent NLPs partition over
        the GPUs, nothing is exchanged) -> `value`.  In the SAME run the knot-sharded path of north_star is measured as well and
        reported beside it (`knot_sharded_allgather`): ONE trajectory whose horizon grows with N (100 knots per GPU), shooting
        intervals sharded contiguously, every step ending with ONE RCCL all-gather of the fused shard buffers + the one-launch
        reassembly of [grad f | jac g | g] in reference order on every rank.  `--shard knots` makes that path the `value`.
The JSON line carries `roofline` (HIP-event timed knot kernel vs the 8 TB/s HBM peak) and `cpu_baseline`
(the CPU oracle timed on the host, rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from hippopt_amd.hipnlp import HipNlp  # noqa: E402
from hippopt_amd.kinodyn_settings import periodic_step_settings, single_step_settings, stairs_settings  # noqa: E402
from hippopt_amd.robot_model import synthetic_ergocub  # noqa: E402
from hippopt_amd.synthetic import make_workload, place_on_step_flanks  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (guides/MI355X_MICROARCH.md)
KNOTS_PER_GPU = 100


def algorithmic_bytes_per_knot(nnz_knot):
    """read x_k and the per-knot parameters, write g_k, the knot's jac values and grad f_k."""
    return 8 * (189 + 79 + 274 + nnz_knot + 189)


def cpu_baseline(settings, model, x, p, budget_s=12.0):
    """The CPU oracle (oracle/kinodyn_oracle.cpp, single thread) on the same workload, bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import Oracle
    orc = Oracle(settings, model)
    orc.eval(x, p)  # warm-up
    t0 = time.perf_counter()
    reps = 0
    while True:
        orc.eval(x, p)
        reps += 1
        el = time.perf_counter() - t0
        if el > budget_s or reps >= 1000:
            break
    out = {"value": settings.horizon_length * reps / el, "unit": "knots/s", "cores": 1, "kind": "port",
           "sample": "%d full callback sets (f, grad f, g, jac g by forward AD) of the N=%d workload in %.1f s, 1 thread"
                     % (reps, settings.horizon_length, el)}
    # beside it: the kernel's own analytic per-knot program run on one CPU thread (the test-only host emulation, tests/hostemu) —
    # hand-derived Jacobians instead of forward AD, i.e. closer to what CasADi's expanded SX graph costs per callback set
    try:
        from hostemu_lib import HostEmu
        emu = HostEmu(settings, model)
        emu.eval(x, p)
        t0 = time.perf_counter()
        r2 = 0
        while time.perf_counter() - t0 < 3.0:
            emu.eval(x, p)
            r2 += 1
        e2 = time.perf_counter() - t0
        out["analytic_port"] = {"value": settings.horizon_length * r2 / e2, "unit": "knots/s", "cores": 1,
                                "sample": "%d callback sets of the analytic knot program on the host (tests/hostemu) in %.1f s, 1 thread" % (r2, e2)}
    except Exception as err:  # noqa: BLE001
        out["analytic_port"] = {"error": str(err)}
    return out


def time_hessian(eng, x_np, knots):
    """ms per evaluation of the exact Hessian of the Lagrangian with device pointers, timed with events on a non-default stream
    (the library maps a null stream pointer to its own stream, which torch events would not see)"""
    import torch
    B = eng.batch
    hn = eng.hess_nnz()
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        xd = torch.tensor(x_np, device="cuda")
        ld = torch.tensor(np.random.RandomState(0).standard_normal((B, eng.m)), device="cuda")
        sd = torch.ones(B, dtype=torch.float64, device="cuda")
        out = torch.zeros((B, hn), dtype=torch.float64, device="cuda")
        reps = 100 if B <= 64 else 20
        for _ in range(5):
            eng.eval_hess_device(xd.data_ptr(), sd.data_ptr(), ld.data_ptr(), out.data_ptr(), stream=stream.cuda_stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            eng.eval_hess_device(xd.data_ptr(), sd.data_ptr(), ld.data_ptr(), out.data_ptr(), stream=stream.cuda_stream)
        e1.record(stream)
    stream.synchronize()
    ms = e0.elapsed_time(e1) / reps
    bytes_knot = 8.0 * (189 + 79 + 274 + hn / (knots / B))
    return {"ms_per_eval": ms, "knots_per_s": knots / (ms * 1e-3), "triplets": int(hn), "algorithmic_bytes_per_knot": bytes_knot,
            "GBps": bytes_knot * knots / (ms * 1e-3) / 1e9, "kernel": "hipnlp_knot_hess_kernel",
            "note": "lower-triangle triplets of sigma hess f + lambda^T hess g, block diagonal by knot; device pointers, no PCIe"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--horizon", type=int, default=KNOTS_PER_GPU)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--workload", choices=["periodic", "single", "stairs"], default="periodic",
                    help="periodic walking (BASELINE metric, default) | single step (final state / periodicity skipped) | stairs (smooth terrain)")
    ap.add_argument("--shard", choices=["knots", "batch"], default="batch")
    ap.add_argument("--also-knot-sharded", action="store_true", help="measure the knot-sharded all-gather path beside `value` even on one GPU (default: whenever WORLD_SIZE > 1)")
    ap.add_argument("--event-stride", type=int, default=16, help="time every n-th launch of the timed region with HIP events")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hessian", action="store_true", help="skip the exact-Hessian figure reported beside the callback quartet")
    ap.add_argument("--force-sharded", action="store_true", help="exercise the sharded path on one GPU (debug)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the engine has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    if args.gpus != world:
        if rank == 0:
            print("warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)

    model = synthetic_ergocub()
    knot_sharded = (world > 1 and args.shard == "knots") or args.force_sharded
    horizon = args.horizon * world if knot_sharded else args.horizon
    settings = {"periodic": periodic_step_settings, "single": single_step_settings, "stairs": stairs_settings}[args.workload](horizon, model)
    seed = 1004 if knot_sharded else 1004 + rank
    x_np, p_np = make_workload(settings, model, batch=args.batch, seed=seed)
    if args.workload == "stairs":   # contact points on the flanks of the bumps: no exp-underflow shortcut in the terrain jets
        place_on_step_flanks(x_np, settings, seed=seed)
    nvar = 4  # a few distinct iterates, cycled: "x changes every call, parameters fixed" 
    rng = np.random.RandomState(5)
    xs = [torch.from_numpy(x_np + 1e-3 * i * rng.standard_normal(x_np.shape)).to(device) for i in range(nvar)]
    stream = torch.cuda.current_stream().cuda_stream

    if knot_sharded:
        from hippopt_amd.sharded import ShardedCallback, hip_shard_backend, hip_shard_info, knot_range
        assert args.batch == 1, "knot sharding evaluates one trajectory"
        kb, ke = knot_range(horizon, world, rank)
        eng = HipNlp(settings, model, batch=1, knot_begin=kb, knot_end=ke, device=local_rank)
        eng.set_params(p_np)
        cb = ShardedCallback(horizon, eng.n, eng.m, eng.nnz, hip_shard_info(eng, kb, ke), hip_shard_backend(eng), device)

        def step(i):
            cb(xs[i % nvar])
        knots_per_step_total = horizon
        local_knots = ke - kb
    else:
        eng = HipNlp(settings, model, batch=args.batch, device=local_rank)
        eng.set_params(p_np)
        f_d = torch.empty(args.batch, dtype=torch.float64, device=device)
        grad_d = torch.empty(args.batch * eng.n, dtype=torch.float64, device=device)
        g_d = torch.empty(args.batch * eng.m, dtype=torch.float64, device=device)
        jac_d = torch.empty(args.batch * eng.nnz, dtype=torch.float64, device=device)

        def step(i):
            eng.eval_device(xs[i % nvar].data_ptr(), f_d.data_ptr(), grad_d.data_ptr(), g_d.data_ptr(), jac_d.data_ptr(), stream=stream)
        knots_per_step_total = horizon * args.batch * world
        local_knots = horizon * args.batch

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    # HIP events around every `stride`-th knot-kernel launch of the timed region: an event record drains the stream and costs
    # microseconds, bracketing EVERY launch would inflate a 100-knot step by two thirds ("Measuring")
    # When a step is exactly ONE kernel launch of ~10 us (latency variant of the engine, no sharded reassembly), an event pair around
    # a single launch would be a quarter of what it measures: the events then bracket RUNS of `stride` consecutive launches
    # (nothing in between) and the run duration is divided by the run length (dispatch gaps inside a run included).
    stride = max(1, args.event_stride)
    single_kernel_step = (not knot_sharded) and eng.kernels_per_eval() == 1
    if single_kernel_step:
        eng.profile_begin_runs(max(1, args.steps // (4 * stride)), stride)
    else:
        eng.profile_begin((args.steps + stride - 1) // stride, stride)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    fence()
    el = time.perf_counter() - t0
    kern_ms, launch_ms, nprof = eng.profile_end()
    timing = ("HIP events around runs of %d consecutive launches / %d (one kernel launch per step)" % (stride, stride)) if single_kernel_step \
        else ("HIP events around every %d-th knot-kernel launch" % stride)
    t = torch.tensor([el], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    el = float(t.item())

    shard_resident = None
    if knot_sharded:
        # the same sharded evaluation WITHOUT the reassembly: every rank leaves its shard's outputs in its own HBM.  Reported
        # beside `value` (which includes the all-gather north_star names), like the PCIe-inclusive rate: never as `value`.
        for i in range(min(args.warmup, 50)):
            cb.compute_shard(xs[i % nvar], *cb.views)
        fence()
        t1 = time.perf_counter()
        for i in range(args.steps):
            cb.compute_shard(xs[i % nvar], *cb.views)
        fence()
        t2 = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=device)
        if world > 1:
            dist.all_reduce(t2, op=dist.ReduceOp.MAX)
        shard_resident = {"knots_per_s": knots_per_step_total * args.steps / float(t2.item()), "ms_per_step": 1e3 * float(t2.item()) / args.steps,
                          "note": "knot shards evaluated, outputs left shard-resident in each rank's HBM (no all-gather / reassembly)"}

    # the knot-sharded all-gather path of north_star, measured in the same run beside a batch-sharded `value`
    ks_extra = None
    if (world > 1 or args.also_knot_sharded) and not knot_sharded and args.batch == 1:
      try:
        from hippopt_amd.sharded import ShardedCallback, hip_shard_backend, hip_shard_info, knot_range
        hz = args.horizon * world
        st2 = {"periodic": periodic_step_settings, "single": single_step_settings, "stairs": stairs_settings}[args.workload](hz, model)
        x2, p2 = make_workload(st2, model, batch=1, seed=1004)   # the SAME trajectory on every rank
        if args.workload == "stairs":
            place_on_step_flanks(x2, st2, seed=1004)
        xs2 = [torch.from_numpy(x2 + 1e-3 * i * np.random.RandomState(5).standard_normal(x2.shape)).to(device) for i in range(nvar)]
        kb, ke = knot_range(hz, world, rank)
        eng2 = HipNlp(st2, model, batch=1, knot_begin=kb, knot_end=ke, device=local_rank)
        eng2.set_params(p2)
        cb2 = ShardedCallback(hz, eng2.n, eng2.m, eng2.nnz, hip_shard_info(eng2, kb, ke), hip_shard_backend(eng2), device)
        ksteps = max(1, min(args.steps, 500))
        for i in range(min(args.warmup, 50)):
            cb2(xs2[i % nvar])
        fence()
        t1 = time.perf_counter()
        for i in range(ksteps):
            cb2(xs2[i % nvar])
        fence()
        t2 = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=device)
        if world > 1:
            dist.all_reduce(t2, op=dist.ReduceOp.MAX)
        ks_extra = {"knots_per_s": hz * ksteps / float(t2.item()), "ms_per_step": 1e3 * float(t2.item()) / ksteps, "horizon": hz,
                    "steps": ksteps, "parallelism": "knot-sharded x%d + all-gather + reassembly" % world,
                    "note": "ONE trajectory of %d knots, %d per GPU: shard evaluation, one all-gather of the fused shard buffers "
                            "(RCCL), one-launch reassembly of [grad | jac | g] in reference order on every rank" % (hz, args.horizon)}
        del cb2, eng2
      except Exception as err:  # noqa: BLE001  (the extra measurement must not take `value` down with it)
        ks_extra = {"error": "%s: %s" % (type(err).__name__, err)}

    if rank == 0:
        d = eng.dims
        nnz_knot = int(d.nnz_knot)
        bytes_knot = algorithmic_bytes_per_knot(nnz_knot)
        bytes_launch = bytes_knot * local_knots
        achieved = bytes_launch / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        traffic = None
        tp = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tp):
            try:
                tj = json.load(open(tp))
                key = "N%d_B%d" % (horizon, args.batch)
                traffic = tj.get(key, {}).get("hbm_bytes_per_launch")
            except Exception:  # noqa: BLE001
                traffic = None
        line = {
            "metric": "NLP callback (f, grad f, g, jac g) throughput, ergoCub-shaped kinodynamic multiple shooting",
            "value": knots_per_step_total * args.steps / el,
            "unit": "knots/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * el / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic (seeded trajectories on a synthetic 23-DoF ergoCub-topology model; no URDF/CasADi in the image)",
            "config": {"workload": {"periodic": "kinodynamic periodic walking, N=%d knots per GPU, batch %d (BASELINE config 4 shape)",
                                    "single": "kinodynamic single step on flat ground, N=%d knots per GPU, batch %d (BASELINE config 3 shape)",
                                    "stairs": "kinodynamic walking on stairs (smooth two-step terrain), N=%d knots per GPU, batch %d (BASELINE config 5 shape)"}[args.workload]
                                   % (args.horizon, args.batch),
                       "horizon": horizon, "batch": args.batch, "knots_per_step": knots_per_step_total,
                       "parallelism": ("knot-sharded x%d + all-gather" % world) if knot_sharded else ("independent trajectories x%d, no collective" % world),
                       "n": int(d.n), "m": int(d.m), "nnz": int(d.nnz), "nnz_per_knot": nnz_knot},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": "hipnlp_knot_kernel", "kernel_ms": kern_ms, "launch_ms": launch_ms,
                         "launches_timed": nprof, "event_stride": stride, "timing": timing, "algorithmic_bytes_per_knot": bytes_knot, "knots_per_launch": local_knots},
        }
        if shard_resident is not None:
            line["shard_resident"] = shard_resident
        if ks_extra is not None:
            line["knot_sharded_allgather"] = ks_extra
        if world == 1 and not knot_sharded:
            # PCIe-inclusive rate of the host-buffer boundary (hipnlp_eval: H2D x, launch, D2H f/grad/g/jac); never `value`
            outs = eng.eval(x_np)
            t1 = time.perf_counter()
            reps = 30
            for i in range(reps):
                eng.eval(x_np, out=outs)   # caller-owned output arrays reused, as IPOPT does
            dt_host = (time.perf_counter() - t1) / reps
            line["pcie_inclusive"] = {"ms_per_call": 1e3 * dt_host, "knots_per_s": horizon * args.batch / dt_host,
                                      "note": "hipnlp_eval with host buffers (pinned staging, one fused D2H copy), all four outputs copied back into caller-owned arrays"}
        if world == 1 and not knot_sharded and not args.no_hessian:
            # beside the callback quartet (never `value`): the exact Hessian of the Lagrangian of the same NLP (hipnlp_eval_hess_device)
            try:
                line["exact_hessian"] = time_hessian(eng, x_np, horizon * args.batch)
            except Exception as err:  # noqa: BLE001  (reported, never fatal to the bench line)
                line["exact_hessian"] = {"error": str(err)}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(settings, model, x_np[0], p_np[0])
            line["cpu_baseline"]["gpu_over_cpu"] = line["value"] / line["cpu_baseline"]["value"]
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
