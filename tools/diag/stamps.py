#!/usr/bin/env python3
"""Diagnostic (never part of the product): per-phase cycle shares of the knot kernel from s_memtime stamps
taken at every workgroup barrier by a -DHIPNLP_STAMPS build (tools/diag/_build/libhipnlp_stamps.so)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
SO = os.path.join(ROOT, "tools", "diag", "_build", "libhipnlp_stamps%s.so" % os.environ.get("STAMPS_VARIANT", ""))


def build():
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DHIPNLP_STAMPS", "-mllvm", "-amdgpu-kernarg-preload-count=16"] + ([] if not os.environ.get("STAMPS_VARIANT") else (["-DHIPNLP_TWOPASS"] if os.environ["STAMPS_VARIANT"] == "twopass" else ["-DHIPNLP_DIAG_SKIP=" + os.environ["STAMPS_VARIANT"]])) + [
                           "-o", SO, os.path.join(ROOT, "hippopt_amd", "csrc", "hipnlp.hip"), os.path.join(ROOT, "hippopt_amd", "csrc", "hipnlp_pose.hip"), os.path.join(ROOT, "hippopt_amd", "csrc", "hipnlp_ipopt.cpp"), "-I", os.path.join(ROOT, "include")])


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
        sys.exit(0)
    from hippopt_amd import hipnlp
    hipnlp._LIB_PATH = SO
    from hippopt_amd.kinodyn_settings import periodic_step_settings, stairs_settings
    from hippopt_amd.robot_model import synthetic_ergocub
    from hippopt_amd.synthetic import make_workload
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    model = synthetic_ergocub()
    stairs = len(sys.argv) > 2 and sys.argv[2] == "stairs"
    st = stairs_settings(100, model) if stairs else periodic_step_settings(100, model)
    x, p = make_workload(st, model, batch, 1004)
    if stairs:
        from hippopt_amd.synthetic import place_on_step_flanks
        place_on_step_flanks(x, st, seed=1)
    eng = hipnlp.HipNlp(st, model, batch=batch, jac_varying_first=os.environ.get('STAMPS_VF') == '1')   # STAMPS_VF=1: the VARY kernels
    eng.set_host_timing(True)
    eng.set_params(p)
    if os.environ.get("STAMPS_DEVICE") == "1":   # the device-pointer path (bench.py's `value`): x resident in HBM
        import torch
        xd = torch.tensor(x, device="cuda")
        outs = [torch.empty(k, dtype=torch.float64, device="cuda") for k in (batch, batch * eng.n, batch * eng.m, batch * eng.nnz)]
        stream = torch.cuda.Stream()
        for _ in range(20):
            eng.eval_device(xd.data_ptr(), *[o.data_ptr() for o in outs], stream=stream.cuda_stream)
        torch.cuda.synchronize()
    else:
        for _ in range(20):
            eng.eval(x)
    out = np.zeros((2 * 100 * batch, 8, 128), np.uint64)    # (room for the two workgroups per knot of a SPLIT launch)
    eng.lib.hipnlp_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
    eng.lib.hipnlp_debug_stamps(eng.h, out.ctypes.data_as(C.c_void_p))
    # SPLIT launches (batch 1, device path): workgroups [0, 100) run the kinematic half of the program, [100, 200) the model-free half.
    # STAMPS_HALF=k / m picks one half (default: k when the launch was split, the whole launch otherwise)
    was_split = batch == 1 and bool(out[100:200, 0, 0].any())
    half = os.environ.get("STAMPS_HALF", "k" if was_split else "")
    if was_split:
        print("SPLIT launch: %s workgroups" % ("kinematic" if half == "k" else "model-free"))
        out = out[:100] if half == "k" else out[100:200]
    else:
        out = out[:100 * batch]
    which = int(os.environ.get('STAMPS_WG', len(out) // 2))
    if which < 0:   # the workgroup with the longest entry -> end
        nb_ = int(out[0, 0, 2])
        oo = out.astype(np.int64)
        which = int(np.argmax(oo[:, :, 8 + 2 * nb_].max(axis=1) - oo[:, :, 0].min(axis=1)))
        print('last workgroup to end:', which)
    blk = out[which].astype(np.int64)  # an interior knot (STAMPS_WG: linear workgroup index)
    waves = [w for w in range(8) if blk[w, 0] != 0]
    t0 = min(blk[w, 0] for w in waves)
    nb = int(blk[waves[0], 2])
    print("interior knot, %d waves, cycles since the first wave entered the kernel (s_memtime); per barrier: arrival of every wave" % len(waves))
    print("loads issued:", [int(blk[w, 5] - t0) for w in waves], " loads arrived:", [int(blk[w, 6] - t0) for w in waves])
    print("staged (after the first barrier):", [int(blk[w, 1] - t0) for w in waves])
    prev = max(blk[w, 1] for w in waves) - t0
    for i in range(nb):
        arr = [int(blk[w, 8 + 2 * i] - t0) for w in waves]
        dep = max(int(blk[w, 9 + 2 * i] - t0) for w in waves)
        print("B%d arrivals %s  last %d (phase %+d)  released %d" % (i, arr, max(arr), max(arr) - prev, dep))
        prev = dep
    # task groups of every wave, in program order (names from the program table: the wave's column of HIPNLP_KNOT_PROGRAM)
    import re
    prog = open(os.path.join(ROOT, "hippopt_amd", "csrc", "knot_body.h")).read()
    prog = prog[prog.index("#define HIPNLP_KNOT_PROGRAM(R, BARRIER)") + len("#define HIPNLP_KNOT_PROGRAM(R, BARRIER)"):]
    prog = re.sub(r"HIPNLP_W[48]\((-?\d+), (-?\d+)\)", (lambda m: m.group(2) if stairs else m.group(1)), prog)
    items = re.findall(r"R\((-?\d+), (-?\d+), (\w+), [^)]*\)|(BARRIER)", prog)
    for w in waves:
        names, b = [], 0
        for w4, w8, fn, bar in items[0:]:
            if bar:
                b += 1
                continue
            if int(w4 if len(waves) == 4 else w8) == w:
                names.append((b, fn))
        n = int(blk[w, 7])
        prev_t = int(blk[w, 1] - t0)
        prev_b = 0
        line = []
        for i in range(min(n, len(names))):
            bb, fn = names[i]
            if bb != prev_b:
                prev_t = int(blk[w, 9 + 2 * (bb - 1)] - t0)   # released from the barrier before this phase
                prev_b = bb
            t = int(blk[w, 32 + i] - t0)
            line.append("%s[%d] %+d" % (fn, bb, t - prev_t))
            prev_t = t
        print("wave %d tasks: %s" % (w, "  ".join(line)))
    end = [int(blk[w, 8 + 2 * nb] - t0) for w in waves]
    print("end (stores issued) %s  (+%d)" % (end, max(end) - prev))
    # the same table as MEDIANS over all workgroups of the launch (one workgroup's stamps move by +-30 % under load)
    oa = out.astype(np.int64)
    t0a = oa[:, waves, 0].min(axis=1)
    ph = []
    prev_a = oa[:, waves, 1].max(axis=1) - t0a
    print("medians over %d workgroups: staged %d" % (len(oa), np.median(prev_a)))
    for i in range(nb):
        last = oa[:, waves, 8 + 2 * i].max(axis=1) - t0a
        rel = oa[:, waves, 9 + 2 * i].max(axis=1) - t0a
        ph.append(int(np.median(last - prev_a)))
        prev_a = rel
    print("  phases (last arrival - previous release): %s   copy-out %d   entry -> end %d" % (
        ph, np.median(oa[:, waves, 8 + 2 * nb].max(axis=1) - t0a - prev_a), np.median(oa[:, waves, 8 + 2 * nb].max(axis=1) - t0a)))
    for w in waves:
        names, b = [], 0
        for w4, w8, fn, bar in items:
            if bar:
                b += 1
                continue
            if int(w4 if len(waves) == 4 else w8) == w:
                names.append((b, fn))
        n = int(blk[w, 7])
        line, tot, prev_b = [], {}, -1
        for i in range(min(n, len(names))):
            bb, fn = names[i]
            if bb != prev_b:
                prev_t = oa[:, w, 9 + 2 * (bb - 1)] if bb > 0 else oa[:, w, 1]
                prev_b = bb
            t = oa[:, w, 32 + i]
            d = int(np.median(t - prev_t))
            tot[bb] = tot.get(bb, 0) + d
            line.append("%s[%d] %d" % (fn, bb, d))
            prev_t = t
        print("  wave %d: %s" % (w, "  ".join(line)))
        print("          per phase: %s" % "  ".join("%d: %d" % kv for kv in sorted(tot.items())))
    if blk[waves[0], 64] != 0:   # -DHIPNLP_TWOPASS: the stamps above are those of the SECOND pass over the knot program
        first = [[int(blk[w, 65 + i] - blk[w, 64]) for i in range(nb)] for w in waves]
        start2 = [int(blk[w, 72]) for w in waves]
        second = [[int(blk[w, 8 + 2 * i]) - start2[j] for i in range(nb)] for j, w in enumerate(waves)]
        print("two passes: arrival at barrier i since the start of the pass, max over waves:  first %s   second %s" % (
            [max(f[i] for f in first) for i in range(nb)], [max(f[i] for f in second) for i in range(nb)]))
    real = [(int(blk[w, 4] - blk[w, 3])) for w in waves]
    cyc = [(int(blk[w, 8 + 2 * nb] - blk[w, 0])) for w in waves]
    print("realtime ticks (100 MHz) per wave %s -> shader clock ~ %.0f MHz" % (real, 100.0 * np.mean(cyc) / max(1.0, np.mean(real))))
    # spread over the grid: entry time of every workgroup relative to the first, and total
    ent = out[:, 0, 0].astype(np.int64)
    endt = out[:, :, 8 + 2 * nb].astype(np.int64).max(axis=1)
    print("grid: first entry -> last entry %d cycles, first entry -> last end %d cycles" % (ent.max() - ent.min(), endt.max() - ent.min()))
    o = out.astype(np.int64)
    # the same on the 100 MHz real-time counter, which all XCDs share (s_memtime is a per-XCD clock: differences ACROSS workgroups of
    # different XCDs mean nothing): entry of every workgroup's first wave and end of its last wave, relative to the first entry
    used = o[:, :len(waves), :]
    r_in, r_out = used[:, :, 3].min(axis=1), used[:, :, 4].max(axis=1)
    t00 = r_in.min()
    print("grid on the real-time counter (10 ns ticks): workgroups enter over %d ticks (median entry %d), end between %d and %d ticks after the first entry; "
          "by XCD (workgroup index mod 8) median entry %s" % (r_in.max() - t00, int(np.median(r_in - t00)), (r_out - t00).min(), (r_out - t00).max(),
                                                              [int(np.median(r_in[j::8] - t00)) for j in range(8)]))
    nkx = len(r_in)
    knot_of = [(xb & 7) * (nkx >> 3) + min(xb & 7, nkx & 7) + (xb >> 3) for xb in range(nkx)]
    late = np.argsort(-r_out)[:5]
    print("last workgroups to end: " + "; ".join("blockIdx %d (knot %d): enters at %d, ends at %d, lasts %d ticks" % (
        int(i), knot_of[int(i)], int(r_in[i] - t00), int(r_out[i] - t00), int(r_out[i] - r_in[i])) for i in late))
    print("workgroup life (ticks): median %d, first knot %d, last knot %d" % (
        int(np.median(r_out - r_in)), int((r_out - r_in)[knot_of.index(0)]), int((r_out - r_in)[knot_of.index(nkx - 1)])))
    if len(waves) == 8:   # publishing wave (7): time from the release of B4 to its arrival at B5, per workgroup
        d = o[:, 7, 8 + 2 * 5] - o[:, 7, 9 + 2 * 4]
        print("wave 7, phase F (release B4 -> arrival B5): median %d  max %d (workgroup %d)" % (np.median(d), d.max(), int(np.argmax(d))))
        e = o[:, :, 8 + 2 * nb].max(axis=1) - o[:, :, 0].min(axis=1)
        print("entry -> end per workgroup: median %d  max %d (workgroup %d)" % (np.median(e), e.max(), int(np.argmax(e))))
        t_first = o[:, :, 0].min()
        print("first entry of the grid -> last end of the grid: %d cycles; last entry %d" % (o[:, :, 8 + 2 * nb].max() - t_first, o[:, :, 0].min(axis=1).max() - t_first))
    if os.environ.get("STAMPS_DEVICE") != "1":
        print("kernel ms:", eng.last_kernel_ms())
