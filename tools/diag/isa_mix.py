#!/usr/bin/env python3
"""Static instruction mix of one kernel of the device assembly, segment by segment between workgroup barriers.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-kernarg-preload-count=16 --cuda-device-only -S -o /tmp/hipnlp.s hippopt_amd/csrc/hipnlp.hip
    python3 tools/diag/isa_mix.py /tmp/hipnlp.s 'hipnlp_knot_kernelILi0ELi4ELb0ELb1E'

The knot program is instantiated once per wave and every task group is one predicated block (or a short loop), so on an interior knot
the STATIC count of a wave's stream is close to what the wave executes: the table says which phase of which wave carries how many fp64
instructions, integer / address instructions, moves and LDS operations (the batch launches are co-bound on VALU issue and the LDS array)."""
import re
import sys

FP = re.compile(r"^v_(add|mul|fma|fmac|max|min|rcp|rsq|sqrt|trig|ldexp|frexp|fract|div|rndne|floor|ceil|cvt|cmp\w*|cndmask)?_?\w*f64")
CLASSES = ("fp64", "int", "mov", "cmp/sel", "lds_rd", "lds_wr", "vmem", "salu", "wait", "other")


def classify(op):
    if op.startswith("v_"):
        if "f64" in op and not op.startswith("v_mov") and not op.startswith("v_cndmask"):
            return "fp64"
        if op.startswith("v_mov") or op.startswith("v_accvgpr") or op.startswith("v_readfirstlane") or op.startswith("v_readlane") or op.startswith("v_writelane"):
            return "mov"
        if op.startswith("v_cmp") or op.startswith("v_cndmask"):
            return "cmp/sel"
        return "int"
    if op.startswith("ds_"):
        return "lds_wr" if ("write" in op or "store" in op) else "lds_rd"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op in ("s_waitcnt", "s_nop", "s_barrier"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().splitlines()
    start = next(i for i, ln in enumerate(lines) if key in ln and ln.rstrip().endswith(":") is False and re.match(r"^_Z\w*:", ln) and key in ln)
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    # the last s_endpgm of the kernel
    j = end
    while True:
        nxt = next((i for i in range(j + 1, min(len(lines), j + 20000)) if lines[i].strip().startswith("s_endpgm") or lines[i].startswith(".Lfunc_end")), None)
        if nxt is None or lines[nxt].startswith(".Lfunc_end"):
            break
        j = nxt
    end = j
    seg, segs = {c: 0 for c in CLASSES}, []
    ops_in = {}
    for ln in lines[start:end + 1]:
        m = re.match(r"^\s+([a-z][a-z0-9_]+)", ln)
        if not m:
            continue
        op = m.group(1)
        c = classify(op)
        seg[c] += 1
        ops_in.setdefault(c, {}).setdefault(op, 0)
        ops_in[c][op] += 1
        if op == "s_barrier":
            segs.append(seg)
            seg = {c: 0 for c in CLASSES}
    segs.append(seg)
    print("%-4s " % "seg" + " ".join("%8s" % c for c in CLASSES) + "   valu")
    tot = {c: 0 for c in CLASSES}
    for i, sg in enumerate(segs):
        valu = sg["fp64"] + sg["int"] + sg["mov"] + sg["cmp/sel"]
        print("%-4d " % i + " ".join("%8d" % sg[c] for c in CLASSES) + "   %d" % valu)
        for c in CLASSES:
            tot[c] += sg[c]
    print("%-4s " % "all" + " ".join("%8d" % tot[c] for c in CLASSES) + "   %d" % (tot["fp64"] + tot["int"] + tot["mov"] + tot["cmp/sel"]))
    for c in ("int", "mov", "cmp/sel"):
        top = sorted(ops_in.get(c, {}).items(), key=lambda kv: -kv[1])[:14]
        print(c + ": " + ", ".join("%s %d" % kv for kv in top))


def tasks():
    """python3 tools/diag/isa_mix.py --tasks /tmp/marks.s <kernel key>: the same classes per TASK GROUP, from an assembly compiled with
    -DHIPNLP_TASK_MARKS (comment markers around every DEV_R group; the markers are scheduling barriers: counts of a marked compile run a
    few instructions above the shipped kernel's)"""
    path, key = sys.argv[2], sys.argv[3]
    lines = open(path).read().splitlines()
    start = next(i for i, ln in enumerate(lines) if re.match(r"^_Z\w*:", ln) and key in ln)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    cur, rows, order = "(outside task groups)", {}, []
    for ln in lines[start:end]:
        t = ln.strip()
        if t.startswith("; TASK_BEGIN"):
            cur = t.split()[2]
            continue
        if t.startswith("; TASK_END"):
            cur = "(outside task groups)"
            continue
        m = re.match(r"^\s+([a-z][a-z0-9_]+)", ln)
        if not m:
            continue
        if cur not in rows:
            rows[cur] = {c: 0 for c in CLASSES}
            order.append(cur)
        rows[cur][classify(m.group(1))] += 1
    print("%-26s " % "task group" + " ".join("%8s" % c for c in CLASSES) + "     valu")
    tot = {c: 0 for c in CLASSES}
    for name in sorted(order, key=lambda n: -(rows[n]["fp64"] + rows[n]["int"] + rows[n]["mov"] + rows[n]["cmp/sel"])):
        sg = rows[name]
        print("%-26s " % name + " ".join("%8d" % sg[c] for c in CLASSES) + "   %6d" % (sg["fp64"] + sg["int"] + sg["mov"] + sg["cmp/sel"]))
        for c in CLASSES:
            tot[c] += sg[c]
    print("%-26s " % "all" + " ".join("%8d" % tot[c] for c in CLASSES) + "   %6d" % (tot["fp64"] + tot["int"] + tot["mov"] + tot["cmp/sel"]))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--tasks":
        tasks()
    else:
        main()
