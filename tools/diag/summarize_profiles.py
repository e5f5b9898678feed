#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (written by tools/diag/profile_round.sh) into the committed summaries under profiles/:
kernel-stats CSVs per workload, PMC means per dispatch, the 8-B-per-lane calibration of FETCH_SIZE / WRITE_SIZE, and
profiles/traffic.json (HBM bytes per launch and VALU wave-instructions per knot that bench.py reports, labelled as looked up)."""
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def newest(pattern):
    """gpurun merges every run into gpurun_out/: keep only the files of the latest run of each pass."""
    files = glob.glob(pattern, recursive=True)
    if not files:
        return []
    t = max(os.path.getmtime(f) for f in files)
    return [f for f in files if os.path.getmtime(f) >= t - 1.0]


def counter_means(d, kernel_substr):
    out = {}
    for f in newest(os.path.join(d, "**", "*counter_collection.csv")):
        acc = {}
        for r in csv.DictReader(open(f)):
            if kernel_substr in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for k, v in acc.items():
            out[k] = sum(v) / len(v)
            out[k + "_n"] = len(v)
    return out


def kernel_stats(d, wanted):
    out = {}
    for f in newest(os.path.join(d, "**", "*kernel_stats.csv")):
        for r in csv.DictReader(open(f)):
            for key, sub in wanted.items():
                if sub in r["Name"]:
                    out[key + "_avg_ns"] = float(r["AverageNs"])
                    out[key + "_calls"] = int(r["Calls"])
                    out[key + "_name"] = r["Name"][:120]
        return out, f
    return out, None


calib = {}
cf = counter_means(os.path.join(src, "calib_fetch"), "copy8")
cw = counter_means(os.path.join(src, "calib_write"), "copy8")
known = float(1 << 30)
calib["fetch_kib_per_launch"] = cf.get("FETCH_SIZE")       # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB
calib["write_kib_per_launch"] = cw.get("WRITE_SIZE")
calib["known_bytes_each"] = known
calib["fetch_correction"] = known / (cf["FETCH_SIZE"] * 1024.0) if cf.get("FETCH_SIZE") else None
calib["write_correction"] = known / (cw["WRITE_SIZE"] * 1024.0) if cw.get("WRITE_SIZE") else None
summary = {"calibration_8B_per_lane": calib, "workloads": {}}
tpath = os.path.join(dst, "traffic.json")
traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
names = sorted({os.path.basename(p)[len("trace_"):] for p in glob.glob(os.path.join(src, "trace_*"))
                if os.path.isdir(p) and not os.path.basename(p).startswith(("trace_hess_", "trace_pose_"))})
for name in names:
    m = re.match(r"(\w+?)_N(\d+)_B(\d+)(_vf|_ccsv)?$", name)   # (_vf: handles with HIPNLP_FLAG_JAC_VARYING_FIRST, VARY kernels; _ccsv: CCS handles with hipnlp_set_constant_jacobian(h, 1))
    N, B = int(m.group(2)), int(m.group(3))
    knots = N * B
    w, f = kernel_stats(os.path.join(src, "trace_" + name), {"knot_kernel": "knot_kernel", "reduce_kernel": "reduce_kernel"})
    if f:
        shutil.copy(f, os.path.join(dst, "%s_kernel_stats_%s.csv" % (tag, name)))
    fe = counter_means(os.path.join(src, "fetch_" + name), "knot_kernel")
    wr = counter_means(os.path.join(src, "write_" + name), "knot_kernel")
    sq = counter_means(os.path.join(src, "sq_" + name), "knot_kernel")
    lds = counter_means(os.path.join(src, "lds_" + name), "knot_kernel")
    lane = counter_means(os.path.join(src, "lane_" + name), "knot_kernel")
    w["FETCH_SIZE_kib"], w["WRITE_SIZE_kib"] = fe.get("FETCH_SIZE"), wr.get("WRITE_SIZE")
    w["lds"] = {k: v for k, v in lds.items() if not k.endswith("_n")}
    w["sq"] = {k: v for k, v in sq.items() if not k.endswith("_n")}
    bj = os.path.join(src, "bench_%s.json" % name)
    if os.path.exists(bj) and os.path.getsize(bj) > 0:
        w["bench"] = json.load(open(bj))
        shutil.copy(bj, os.path.join(dst, "%s_bench_%s.json" % (tag, name)))
        w["algorithmic_bytes_per_launch"] = w["bench"]["roofline"]["algorithmic_bytes_per_knot"] * knots
    ent = {"round": tag}
    if fe.get("FETCH_SIZE") is not None and wr.get("WRITE_SIZE") is not None and calib["fetch_correction"]:
        hbm = fe["FETCH_SIZE"] * 1024.0 * calib["fetch_correction"] + wr["WRITE_SIZE"] * 1024.0 * calib["write_correction"]
        w["hbm_bytes_per_launch_corrected"] = hbm
        ent.update({"hbm_bytes_per_launch": hbm, "fetch_kib": fe["FETCH_SIZE"], "write_kib": wr["WRITE_SIZE"],
                    "fetch_correction": calib["fetch_correction"], "write_correction": calib["write_correction"]})
    if sq.get("SQ_INSTS_VALU"):
        ent["valu_wave_insts_per_knot"] = sq["SQ_INSTS_VALU"] / knots
        w["valu_wave_insts_per_knot"] = ent["valu_wave_insts_per_knot"]
        w["salu_wave_insts_per_knot"] = sq.get("SQ_INSTS_SALU", 0.0) / knots
        w["lds_wave_insts_per_knot"] = sq.get("SQ_INSTS_LDS", 0.0) / knots
        if sq.get("SQ_WAVE_CYCLES"):
            w["wait_fraction_of_wave_cycles"] = sq.get("SQ_WAIT_ANY", 0.0) / sq["SQ_WAVE_CYCLES"]
    if lds.get("SQ_LDS_IDX_ACTIVE"):
        ent["lds_array_cycles_per_knot"] = lds["SQ_LDS_IDX_ACTIVE"] / knots
        w["lds_array_cycles_per_knot"] = ent["lds_array_cycles_per_knot"]
        w["lds_bank_conflict_cycles_per_knot"] = lds.get("SQ_LDS_BANK_CONFLICT", 0.0) / knots
        if lds.get("SQ_BUSY_CU_CYCLES"):
            w["lds_array_busy_fraction_of_cu_cycles"] = lds["SQ_LDS_IDX_ACTIVE"] / lds["SQ_BUSY_CU_CYCLES"]
    if lane.get("SQ_ACTIVE_INST_VALU") and lane.get("SQ_THREAD_CYCLES_VALU"):
        # live lanes per issued VALU cycle: thread-cycles / (instruction-cycles x 64 lanes)
        w["lane"] = {k: v for k, v in lane.items() if not k.endswith("_n")}
        ent["valu_lane_utilisation"] = w["valu_lane_utilisation"] = lane["SQ_THREAD_CYCLES_VALU"] / (lane["SQ_ACTIVE_INST_VALU"] * 64.0)
        if lane.get("SQ_WAVE_CYCLES"):
            w["valu_active_fraction_of_wave_cycles"] = lane["SQ_ACTIVE_INST_VALU"] / lane["SQ_WAVE_CYCLES"]
            w["lds_issue_stall_fraction_of_wave_cycles"] = lane.get("SQ_WAIT_INST_LDS", 0.0) / lane["SQ_WAVE_CYCLES"]
    if len(ent) > 1:
        traffic[name] = ent
    summary["workloads"][name] = w
# exact Hessian and pose finder: one (kernel, N, batch) per trace directory (trace_hess_<terrain>_N<N>_B<B>, trace_pose_<what>_B<B>)
summary["exact_hessian"], summary["pose_finder"] = {}, {}
for d in sorted(glob.glob(os.path.join(src, "trace_hess_*")) + glob.glob(os.path.join(src, "trace_pose_*"))):
    if not os.path.isdir(d):
        continue
    name = os.path.basename(d)[len("trace_"):]
    kern = "knot_hess_kernel" if name.startswith("hess_") else ("pose_hess_kernel" if "hessian" in name else "pose_kernel")
    h, f = kernel_stats(d, {"kernel": kern})
    if f:
        shutil.copy(f, os.path.join(dst, "%s_kernel_stats_%s.csv" % (tag, name)))
    bj = os.path.join(src, "bench_%s.jsonl" % name)
    if os.path.exists(bj):
        lines = [l for l in open(bj) if l.startswith("{")]
        h["bench"] = [json.loads(l) for l in lines]
    fe = counter_means(os.path.join(src, "fetch_" + name), kern)
    wr = counter_means(os.path.join(src, "write_" + name), kern)
    lane = counter_means(os.path.join(src, "lane_" + name), kern)
    ent = {"round": tag, "kernel": kern}
    if h.get("kernel_avg_ns"):
        ent["kernel_avg_ns"] = h["kernel_avg_ns"]
    if fe.get("FETCH_SIZE") is not None and wr.get("WRITE_SIZE") is not None and calib["fetch_correction"]:
        ent["hbm_bytes_per_launch"] = h["hbm_bytes_per_launch_corrected"] = fe["FETCH_SIZE"] * 1024.0 * calib["fetch_correction"] + wr["WRITE_SIZE"] * 1024.0 * calib["write_correction"]
    if lane.get("SQ_ACTIVE_INST_VALU") and lane.get("SQ_THREAD_CYCLES_VALU"):
        ent["valu_lane_utilisation"] = h["valu_lane_utilisation"] = lane["SQ_THREAD_CYCLES_VALU"] / (lane["SQ_ACTIVE_INST_VALU"] * 64.0)
        h["lane"] = {k: v for k, v in lane.items() if not k.endswith("_n")}
    # per unit (knot-Hessian / pose) and against the CU-busy cycles: which unit the batch launch keeps busy
    mu = re.search(r"_N(\d+)_B(\d+)$", name) or re.search(r"_B(\d+)$", name)
    units = (int(mu.group(1)) * int(mu.group(2)) if mu.re.groups == 2 else int(mu.group(1))) if mu else None
    ldsc = counter_means(os.path.join(src, "lds_" + name), kern)
    if units and lane.get("SQ_INSTS_VALU"):
        ent["valu_wave_insts_per_unit"] = h["valu_wave_insts_per_unit"] = lane["SQ_INSTS_VALU"] / units
        h["lds_wave_insts_per_unit"] = lane.get("SQ_INSTS_LDS", 0.0) / units
        if lane.get("SQ_WAVE_CYCLES"):
            h["wait_fraction_of_wave_cycles"] = lane.get("SQ_WAIT_ANY", 0.0) / lane["SQ_WAVE_CYCLES"]
    if units and ldsc.get("SQ_LDS_IDX_ACTIVE"):
        ent["lds_array_cycles_per_unit"] = h["lds_array_cycles_per_unit"] = ldsc["SQ_LDS_IDX_ACTIVE"] / units
        if ldsc.get("SQ_BUSY_CU_CYCLES"):
            h["cu_busy_cycles_per_unit"] = ldsc["SQ_BUSY_CU_CYCLES"] / units
            h["lds_array_busy_fraction_of_cu_cycles"] = ldsc["SQ_LDS_IDX_ACTIVE"] / ldsc["SQ_BUSY_CU_CYCLES"]
            if lane.get("SQ_INSTS_VALU"):   # fp64 VALU: one wave-instruction holds a SIMD's issue slot for 4 cycles, 4 SIMDs per CU
                h["valu_issue_busy_fraction_of_cu_cycles"] = lane["SQ_INSTS_VALU"] / ldsc["SQ_BUSY_CU_CYCLES"]
        h["lds"] = {k: v for k, v in ldsc.items() if not k.endswith("_n")}
    traffic[name] = ent
    summary["exact_hessian" if name.startswith("hess_") else "pose_finder"][name] = h
json.dump(summary, open(os.path.join(dst, "%s_summary.json" % tag), "w"), indent=1)
json.dump(traffic, open(tpath, "w"), indent=1)
print(json.dumps(calib, indent=1))
for k, v in summary["workloads"].items():
    line = "%-22s knot kernel %.1f ns x %s" % (k, v.get("knot_kernel_avg_ns", float("nan")), v.get("knot_kernel_calls"))
    if "bench" in v:
        b = v["bench"]
        line += " | bench %.3g knots/s kernel_ms %.5f frac %.4f" % (b["value"], b["roofline"]["kernel_ms"], b["roofline"]["frac"])
    if "hbm_bytes_per_launch_corrected" in v:
        line += " | HBM %.3g B vs algorithmic %.3g B" % (v["hbm_bytes_per_launch_corrected"], v.get("algorithmic_bytes_per_launch", float("nan")))
    if "valu_wave_insts_per_knot" in v:
        line += " | VALU/knot %.0f SALU %.0f LDS %.0f wait %.2f" % (v["valu_wave_insts_per_knot"], v["salu_wave_insts_per_knot"], v["lds_wave_insts_per_knot"], v.get("wait_fraction_of_wave_cycles", float("nan")))
    print(line)
for group in ("exact_hessian", "pose_finder"):
    for name, h in summary[group].items():
        print(group, name, {k: v for k, v in h.items() if k not in ("bench", "lane")})
