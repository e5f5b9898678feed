"""Comparison of two Hessians given as lower-triangle triplets (pattern supersets allowed: entries absent on one side are zeros)."""
import numpy as np


def triplets_to_dict(rows, cols, vals):
    d = {}
    for r, c, v in zip(np.asarray(rows).tolist(), np.asarray(cols).tolist(), np.asarray(vals).tolist()):
        assert r >= c, "entry above the diagonal"
        assert (r, c) not in d, "duplicate entry (%d, %d)" % (r, c)
        d[(r, c)] = v
    return d


def hess_mismatch(mine, ref, diag_scaled=False):
    """max over entries of |a - b| / max(1, |b|), entries present on one side only compared with 0.
    diag_scaled: the denominator is max(1, |b|, sqrt(|b_rr b_cc|)) — the scale of entry (r, c) of a symmetric matrix is that of its two
    diagonal entries.  Smooth-terrain tests only: on the flank of a bump the (p, p) block of a contact point has entries of 1e4 - 1e8
    made of terms that cancel, and a small off-diagonal entry inside such a block carries their rounding noise (seed 3300: entry
    (p_y, p_x) = -1.13 beside a diagonal of -8.1e4 differs from the oracle's forward-over-forward AD by 1.3e-9, two evaluation orders
    of the kernel's own arithmetic agree with each other to 2e-10 and lie on the same side of the oracle)."""
    worst, where = 0.0, None
    for k in set(mine) | set(ref):
        a, b = mine.get(k, 0.0), ref.get(k, 0.0)
        scale = max(1.0, abs(b))
        if diag_scaled:
            scale = max(scale, np.sqrt(abs(ref.get((k[0], k[0]), 0.0)) * abs(ref.get((k[1], k[1]), 0.0))))
        e = abs(a - b) / scale if np.isfinite(a) else np.inf
        if e > worst:
            worst, where = e, k
    return worst, where
