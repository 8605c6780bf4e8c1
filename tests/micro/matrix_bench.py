"""Fill time of config 2's shape (10 000 pairs, 2 kb x ~2 kb, band radius 200, B_LOCAL) under four scorings:
match / mismatch (packed x4), a non-uniform 4 x 4 integer matrix (packed, matrix form), the same through the generic
kernel, and dyadic scores.  Usage: python tests/micro/matrix_bench.py [pairs] [reps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from biseqt_amd import _pwlib as W            # noqa: E402
from biseqt_amd import synth                  # noqa: E402
from biseqt_amd.batch import BatchAligner     # noqa: E402

BLASTISH = [[1, -3, -2, -3], [-3, 1, -3, -2], [-2, -3, 1, -3], [-3, -2, -3, 1]]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    origins, mutants = synth.pair_batch(2, n, 2000)
    pairs = list(zip(origins, mutants))
    base = dict(alnmode=1, alntype=1, alphabet_len=4, diag_range=(-200, 200), go_score=-5, ge_score=-2)
    runs = [('match/mismatch 1/-3', dict(match_score=1, mismatch_score=-3), 0),
            ('4x4 matrix, packed', dict(subst_scores=BLASTISH), 0),
            ('4x4 matrix, generic', dict(subst_scores=BLASTISH), W.PW_FLAG_FORCE_GENERIC),
            ('dyadic 0.25/-0.75', dict(match_score=0.25, mismatch_score=-0.75, go_score=-1.25, ge_score=-0.5), 0),
            ('dyadic, forced f64', dict(match_score=0.25, mismatch_score=-0.75, go_score=-1.25, ge_score=-0.5), W.PW_FLAG_FORCE_F64)]
    for title, skw, flags in runs:
        kw = dict(base)
        kw.update(skw)
        with BatchAligner(pairs, flags=flags | W.PW_FLAG_PROFILE, **kw) as b:
            ts, tr = [], []
            for _ in range(reps):
                b.solve(); b.traceback(); b.sync()
                ts.append(b.fill_ms()); tr.append(b.trace_ms())
            cells = b.cells
            print('%-24s %-34s fill %7.3f ms (best %7.3f) = %7.1f GCUPS   traceback %.3f ms' %
                  (title, b.kernel_name, float(np.median(ts)), min(ts), cells / min(ts) / 1e6, float(np.median(tr))), flush=True)


if __name__ == '__main__':
    main()
