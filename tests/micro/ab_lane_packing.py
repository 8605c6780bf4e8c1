"""Config 2's shape (10 000 pairs, 2 kb, band radius 200, B_LOCAL) on other lane layouts of the packed kernel: one pair per
wavefront at 8 diagonals per lane keeps 401 of 512 slots busy; 3 pairs side by side at 20 per lane would keep 94 % busy, 4 at 28
per lane 89.5 % (PWLIB_PACKED_BK=<bk>s forces the lane-packed layout).  Alternating launches on the same pairs.

    python tests/micro/ab_lane_packing.py [rounds]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from biseqt_amd import _pwlib as W            # noqa: E402
from biseqt_amd import synth                  # noqa: E402
from biseqt_amd.batch import BatchAligner     # noqa: E402


def make(pairs, env):
    for k in ('PWLIB_PACKED_BK', 'PWLIB_SIMPLE_AS_MATRIX'):
        os.environ.pop(k, None)
    os.environ.update(env)
    b = BatchAligner(pairs, flags=W.PW_FLAG_PROFILE, alnmode=1, alntype=1, diag_range=(-200, 200), alphabet_len=4,
                     match_score=1, mismatch_score=-3, go_score=-5, ge_score=-2)
    b.__enter__()
    return b


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    origins, mutants = synth.pair_batch(2, 10000, 2000)
    pairs = list(zip(origins, mutants))
    variants = [('default', {}), ('plain 8', {'PWLIB_SIMPLE_AS_MATRIX': '0'}),
                ('plain 20s', {'PWLIB_PACKED_BK': '20s', 'PWLIB_SIMPLE_AS_MATRIX': '0'}), ('matrix 20s', {'PWLIB_PACKED_BK': '20s', 'PWLIB_SIMPLE_AS_MATRIX': '1'}),
                ('plain 28s', {'PWLIB_PACKED_BK': '28s', 'PWLIB_SIMPLE_AS_MATRIX': '0'}), ('matrix 28s', {'PWLIB_PACKED_BK': '28s', 'PWLIB_SIMPLE_AS_MATRIX': '1'}),
                ('plain 16s', {'PWLIB_PACKED_BK': '16s', 'PWLIB_SIMPLE_AS_MATRIX': '0'})]
    bs = [(t, make(pairs, e), []) for t, e in variants]
    for r in range(rounds + 1):
        for t, b, ts in bs:
            b.solve(); b.sync()
            if r:
                ts.append(b.fill_ms())
    ref = bs[0][1].results()
    for t, b, ts in bs:
        print('%-12s %-34s fill %7.3f ms (best %7.3f)   records equal to the default: %s'
              % (t, b.kernel_name, float(np.median(ts)), min(ts), bool(np.array_equal(b.results(), ref))), flush=True)
        b.__exit__(None, None, None)


if __name__ == '__main__':
    main()
