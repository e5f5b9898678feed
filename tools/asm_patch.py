#!/usr/bin/env python3
"""Build step (used by __graft_entry__.build): splits the two-address LDS reads the compiler forms into single-address ones in the
gfx950 assembly of the knot kernels.

`ds_read2_b64 v[a:a+3], vaddr offset0:X offset1:Y` reads two doubles with ONE wave instruction, but the LDS array serves it at half
the rate of two `ds_read_b64` (8.2 against 5.5 cycles per pair with sixteen waves on the CU, tools/diag/lds_rate_probe.hip,
profiles/r02_lds_rate_probe.txt) — and at four workgroups per CU the LDS array is the busiest unit of the callback kernels (70 % of
the CU cycles, 40 % of them on these pairs).  The compiler forms them in two places (the IR load-store vectoriser and the machine
load-store optimiser) and offers no switch for either on LDS, so the split is done on its output:

    ds_read2_b64     v[a:a+3], vA offset0:X offset1:Y   ->   ds_read_b64 v[a:a+1], vA offset:8X  ;  ds_read_b64 v[a+2:a+3], vA offset:8Y
    ds_read2st64_b64 v[a:a+3], vA offset0:X offset1:Y   ->   the same with offsets 512X, 512Y

Waits stay correct: LDS operations return in order and `s_waitcnt lgkmcnt(N)` means "at most N outstanding"; with more, shorter
operations in flight every existing wait covers at least the operations it covered before.  Where the destination contains the
address register, the half that overwrites it is issued last (an LDS read takes its address at issue).

usage: asm_patch.py in.s out.s [kernel-name-substring ...]      (default: every function)"""
import re
import sys

PAT = re.compile(r"^(\s*)ds_read2(st64)?_b64\s+v\[(\d+):(\d+)\],\s*v(\d+)((?:\s+offset[01]:\d+)*)\s*$")


def patch(text, only=()):
    out, n, skipped = [], 0, 0
    active = not only
    for line in text.split("\n"):
        if only and line[:1] not in (" ", "\t", ".", ";", "") and ":" in line.split(";")[0]:   # a function label
            active = any(k in line for k in only)
        m = PAT.match(line) if active else None
        if not m:
            out.append(line)
            continue
        ind, st64, lo, hi, addr, offs = m.group(1), m.group(2), int(m.group(3)), int(m.group(4)), int(m.group(5)), m.group(6)
        if hi - lo != 3:
            out.append(line)
            skipped += 1
            continue
        o = {"0": 0, "1": 0}
        for k, v in re.findall(r"offset([01]):(\d+)", offs):
            o[k] = int(v)
        unit = 512 if st64 else 8
        # (the address register is read at issue: the half whose destination contains it goes last)
        order = ((1, "1"), (0, "0")) if lo <= addr <= lo + 1 else ((0, "0"), (1, "1"))
        for half, key in order:
            off = o[key] * unit
            out.append("%sds_read_b64 v[%d:%d], v%d%s" % (ind, lo + 2 * half, lo + 2 * half + 1, addr, (" offset:%d" % off) if off else ""))
        n += 1
    return "\n".join(out), n, skipped


if __name__ == "__main__":
    src, dst = sys.argv[1], sys.argv[2]
    text, n, skipped = patch(open(src).read(), tuple(sys.argv[3:]))
    open(dst, "w").write(text)
    print("asm_patch: %d ds_read2_b64 split, %d left" % (n, skipped))
