"""Comparison of two Hessians given as lower-triangle triplets (pattern supersets allowed: entries absent on one side are zeros)."""
import numpy as np


def triplets_to_dict(rows, cols, vals):
    d = {}
    for r, c, v in zip(np.asarray(rows).tolist(), np.asarray(cols).tolist(), np.asarray(vals).tolist()):
        assert r >= c, "entry above the diagonal"
        assert (r, c) not in d, "duplicate entry (%d, %d)" % (r, c)
        d[(r, c)] = v
    return d


def hess_mismatch(mine, ref):
    """max over entries of |a - b| / max(1, |b|), entries present on one side only compared with 0"""
    worst, where = 0.0, None
    for k in set(mine) | set(ref):
        a, b = mine.get(k, 0.0), ref.get(k, 0.0)
        e = abs(a - b) / max(1.0, abs(b)) if np.isfinite(a) else np.inf
        if e > worst:
            worst, where = e, k
    return worst, where
