"""One caller, several shard handles of ONE card (hipnlp_multi_create): what the host side of the path costs.

The box has one card, so the shards share its link: nothing here can be faster than the plain handle.  What the numbers say is how much
the caller's side ADDS per shard — launches and waits from one thread, or one launching thread per shard — which is what stands between
n links and n times the bytes per second.

usage: python tools/diag/multi_probe.py [horizon] [calls]"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from hippopt_amd.hipnlp import HipNlp, pin_to_device_numa_node  # noqa: E402
from hippopt_amd.kinodyn_settings import periodic_step_settings  # noqa: E402
from hippopt_amd.robot_model import synthetic_ergocub  # noqa: E402
from hippopt_amd.synthetic import make_workload  # noqa: E402


def timed(eng, xs, out, calls, want=("f", "grad", "g", "jac")):
    for i in range(20):
        eng.eval(xs[i % len(xs)], out=out, want=want)
    best = 1e30
    for _ in range(4):
        t0 = time.perf_counter()
        for i in range(calls):
            eng.eval(xs[i % len(xs)], out=out, want=want)
        best = min(best, (time.perf_counter() - t0) / calls * 1e6)
    return best


def main():
    horizon = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    calls = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    model = synthetic_ergocub()
    pin = pin_to_device_numa_node(0)
    st = periodic_step_settings(horizon, model)
    x, p = make_workload(st, model, batch=1, seed=1004)
    rng = np.random.RandomState(0)
    xs = [x + 1e-4 * rng.standard_normal(x.shape) for _ in range(8)]
    res = {"horizon": horizon, "pinned_to_node": pin, "rows": []}
    lam = rng.standard_normal((1, 1))
    for shards, threads in [(0, False), (1, False), (2, False), (2, True), (4, False), (4, True), (8, False), (8, True)]:
        eng = HipNlp(st, model, detect_simple_bounds=True, jac_varying_first=True, devices=None if shards == 0 else [0] * shards)
        eng.set_params(p)
        if threads:
            eng.set_threads(True)
        out = (np.empty(1), np.empty((1, eng.n)), np.empty((1, eng.m)), np.empty((1, eng.nnz)))
        row = {"shards": shards, "threads": threads}
        row["all_us"] = round(timed(eng, xs, out, calls), 2)
        row["f_g_us"] = round(timed(eng, xs, out, calls, want=("f", "g")), 2)
        eng.eval(xs[0], out=out)
        row["host_breakdown_us"] = [round(v, 2) for v in eng.host_breakdown()]
        if shards:
            row["shard_us"] = np.round(eng.multi_breakdown(), 1).tolist()
        lam = rng.standard_normal((1, eng.m))
        hess = np.empty((1, eng.hess_nnz()))
        for _ in range(30):
            eng.eval_hess(xs[0], 1.0, lam, out=hess)
        t0 = time.perf_counter()
        for i in range(calls):
            eng.eval_hess(xs[i % len(xs)], 1.0, lam, out=hess)
        row["hess_us"] = round((time.perf_counter() - t0) / calls * 1e6, 2)
        res["rows"].append(row)
        print(json.dumps(row), flush=True)
        eng.close()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
