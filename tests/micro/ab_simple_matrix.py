"""A/B inside one process: match / mismatch scoring on the packed kernel's own form (xor, min, multiply-add) against the same
scores fed through its matrix form (PWLIB_SIMPLE_AS_MATRIX=1: one v_perm_b32), alternating launches on the same pairs.

    python tests/micro/ab_simple_matrix.py [rounds]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from biseqt_amd import _pwlib as W            # noqa: E402
from biseqt_amd import synth                  # noqa: E402
from biseqt_amd.batch import BatchAligner     # noqa: E402


def make(pairs, env, **kw):
    os.environ['PWLIB_SIMPLE_AS_MATRIX'] = env
    b = BatchAligner(pairs, flags=W.PW_FLAG_PROFILE, alphabet_len=4, match_score=1, mismatch_score=-3, go_score=-5, ge_score=-2, **kw)
    b.__enter__()
    return b


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    shapes = [('config 2: 10000 x 2 kb, radius 200, B_LOCAL', 10000, 2000, dict(alnmode=1, alntype=1, diag_range=(-200, 200))),
              ('3000 x 2 kb, radius 400, B_LOCAL', 3000, 2000, dict(alnmode=1, alntype=1, diag_range=(-400, 400))),
              ('10000 x 2 kb, radius 200, B_OVERLAP', 10000, 2000, dict(alnmode=1, alntype=2, diag_range=(-200, 200))),
              ('20000 x 300, radius 10, B_LOCAL (lane-packed)', 20000, 300, dict(alnmode=1, alntype=1, diag_range=(-10, 10))),
              ('5000 x 1 kb standard GLOBAL', 5000, 1000, dict(alnmode=0, alntype=0))]
    for title, n, length, kw in shapes:
        origins, mutants = synth.pair_batch(2, n, length)
        pairs = list(zip(origins, mutants))
        A, B = make(pairs, '0', **kw), make(pairs, '1', **kw)
        ta, tb = [], []
        for r in range(rounds + 2):
            for b, ts in ((A, ta), (B, tb)) if r % 2 == 0 else ((B, tb), (A, ta)):
                b.solve(); b.sync()
                if r >= 2:
                    ts.append(b.fill_ms())
        ra, rb = A.results(), B.results()
        same = bool(np.array_equal(ra, rb))
        print('%-48s %-28s %7.3f ms (best %7.3f) | %-28s %7.3f ms (best %7.3f) | matrix form %+5.1f %%  records equal: %s' %
              (title, A.kernel_name, float(np.median(ta)), min(ta), B.kernel_name, float(np.median(tb)), min(tb),
               100.0 * (float(np.median(tb)) / float(np.median(ta)) - 1.0), same), flush=True)
        A.__exit__(None, None, None); B.__exit__(None, None, None)


if __name__ == '__main__':
    main()
