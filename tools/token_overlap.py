#!/usr/bin/env python3
"""Self-check against copying: share of a file's tokens that lie in runs of >= K tokens also present in another file
(Python tokenizer; comments, strings' content and layout ignored).  usage: tools/token_overlap.py mine.py theirs.py [K=12]"""
import io
import sys
import tokenize


def tokens(path):
    out = []
    with open(path, "rb") as f:
        for tok in tokenize.tokenize(f.readline):
            if tok.type in (tokenize.COMMENT, tokenize.NL, tokenize.NEWLINE, tokenize.INDENT, tokenize.DEDENT, tokenize.ENCODING, tokenize.ENDMARKER):
                continue
            if tok.type == tokenize.STRING and tok.string.lstrip("rbfuRBFU").startswith(('"""', "'''")):
                continue   # docstrings
            out.append(tok.string)
    return out


def overlap(mine, theirs, k=12):
    a, b = tokens(mine), tokens(theirs)
    grams = {}
    for i in range(len(b) - k + 1):
        grams.setdefault(tuple(b[i:i + k]), i)
    covered = [False] * len(a)
    for i in range(len(a) - k + 1):
        if tuple(a[i:i + k]) in grams:
            for j in range(i, i + k):
                covered[j] = True
    return sum(covered) / max(1, len(a)), len(a)


if __name__ == "__main__":
    k = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    frac, n = overlap(sys.argv[1], sys.argv[2], k)
    print("%.1f %% of %d tokens in shared runs of >= %d tokens" % (100 * frac, n, k))
