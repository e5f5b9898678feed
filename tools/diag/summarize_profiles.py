#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (written by tools/diag/profile_round.sh) into the committed summaries under profiles/:
kernel stats CSVs, PMC means per dispatch, the 8-B-per-lane calibration of FETCH_SIZE / WRITE_SIZE, and
profiles/traffic.json (HBM bytes per launch that bench.py reports as roofline.traffic)."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def newest(pattern):
    """gpurun merges every run into gpurun_out/: keep only the files of the latest run of each pass."""
    files = glob.glob(pattern)
    if not files:
        return []
    t = max(os.path.getmtime(f) for f in files)
    return [f for f in files if os.path.getmtime(f) >= t - 1.0]


def counter_means(d, kernel_substr):
    out = {}
    for f in newest(os.path.join(d, "*", "*counter_collection.csv")):
        acc = {}
        for r in csv.DictReader(open(f)):
            if kernel_substr in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for k, v in acc.items():
            out[k] = sum(v) / len(v)
            out[k + "_n"] = len(v)
    return out


calib = {}
cf = counter_means(os.path.join(src, "calib_fetch"), "copy8")
cw = counter_means(os.path.join(src, "calib_write"), "copy8")
known = float(1 << 30)
# rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB
calib["fetch_kib_per_launch"] = cf.get("FETCH_SIZE")
calib["write_kib_per_launch"] = cw.get("WRITE_SIZE")
calib["known_bytes_each"] = known
calib["fetch_correction"] = known / (cf["FETCH_SIZE"] * 1024.0) if cf.get("FETCH_SIZE") else None
calib["write_correction"] = known / (cw["WRITE_SIZE"] * 1024.0) if cw.get("WRITE_SIZE") else None
summary = {"calibration_8B_per_lane": calib, "workloads": {}}
traffic = {}
for B in (1, 64, 1024):
    w = {}
    for f in newest(os.path.join(src, "trace_B%d" % B, "*", "*kernel_stats.csv")):
        shutil.copy(f, os.path.join(dst, "%s_kernel_stats_N100_B%d.csv" % (tag, B)))
        for r in csv.DictReader(open(f)):
            if "knot_kernel" in r["Name"]:
                w["knot_kernel_avg_ns"] = float(r["AverageNs"])
                w["knot_kernel_calls"] = int(r["Calls"])
            if "reduce_kernel" in r["Name"]:
                w["reduce_kernel_avg_ns"] = float(r["AverageNs"])
    fe = counter_means(os.path.join(src, "fetch_B%d" % B), "knot_kernel")
    wr = counter_means(os.path.join(src, "write_B%d" % B), "knot_kernel")
    sq = counter_means(os.path.join(src, "sq_B%d" % B), "knot_kernel")
    w["FETCH_SIZE_kib"] = fe.get("FETCH_SIZE")
    w["WRITE_SIZE_kib"] = wr.get("WRITE_SIZE")
    w["sq"] = {k: v for k, v in sq.items() if not k.endswith("_n")}
    if fe.get("FETCH_SIZE") is not None and wr.get("WRITE_SIZE") is not None and calib["fetch_correction"]:
        hbm = fe["FETCH_SIZE"] * 1024.0 * calib["fetch_correction"] + wr["WRITE_SIZE"] * 1024.0 * calib["write_correction"]
        w["hbm_bytes_per_launch_corrected"] = hbm
        w["algorithmic_bytes_per_launch"] = 16936 * 100 * B
        traffic["N100_B%d" % B] = {"hbm_bytes_per_launch": hbm, "fetch_kib": fe["FETCH_SIZE"], "write_kib": wr["WRITE_SIZE"],
                                   "fetch_correction": calib["fetch_correction"], "write_correction": calib["write_correction"], "round": tag}
    bj = os.path.join(src, "bench_B%d.json" % B)
    if os.path.exists(bj) and os.path.getsize(bj) > 0:
        w["bench"] = json.load(open(bj))
        shutil.copy(bj, os.path.join(dst, "%s_bench_N100_B%d.json" % (tag, B)))
    summary["workloads"]["N100_B%d" % B] = w
for f in newest(os.path.join(src, "trace_hess", "*", "*kernel_stats.csv")):
    shutil.copy(f, os.path.join(dst, "%s_hess_kernel_stats.csv" % tag))
hb = os.path.join(src, "hess_bench.jsonl")
if os.path.exists(hb):
    lines = [l for l in open(hb) if l.startswith("{")]
    open(os.path.join(dst, "%s_hess_bench.jsonl" % tag), "w").writelines(lines)
    summary["exact_hessian"] = [json.loads(l) for l in lines]
json.dump(summary, open(os.path.join(dst, "%s_summary.json" % tag), "w"), indent=1)
json.dump(traffic, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk not in ("bench", "sq")} for k, v in summary["workloads"].items()}, indent=1))
print(json.dumps(calib, indent=1))
for k, v in summary["workloads"].items():
    if "bench" in v:
        b = v["bench"]
        print(k, "knots/s %.3g" % b["value"], "kernel_ms %.4f" % b["roofline"]["kernel_ms"], "frac %.4f" % b["roofline"]["frac"])
