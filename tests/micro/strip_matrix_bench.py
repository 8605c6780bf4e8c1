"""One wide standard-mode pair with a 4 x 4 substitution matrix: the strips (byte rows, round 3) against the tiled kernel such
pairs ran on before.  Usage: python tests/micro/strip_matrix_bench.py [n ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from biseqt_amd import _pwlib as W            # noqa: E402
from biseqt_amd import synth                  # noqa: E402
from biseqt_amd.batch import BatchAligner     # noqa: E402

BLASTISH = [[1, -3, -2, -3], [-3, 1, -3, -2], [-2, -3, 1, -3], [-3, -2, -3, 1]]


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [4000, 8000, 20000]
    rng = synth.rng_for(11)
    for n in sizes:
        o = synth.rand_seqs(rng, 1, n)[0]
        m = synth.mutate(rng, o, 0.07, 0.015, 0.5)
        out = []
        recs = []
        for title, flags in (('planner', 0), ('tiled', W.PW_FLAG_FORCE_TILED)):
            with BatchAligner([(o, m)], alnmode=0, alntype=1, alphabet_len=4, subst_scores=BLASTISH, go_score=-5, ge_score=-2,
                              flags=flags | W.PW_FLAG_PROFILE) as b:
                ts = []
                for _ in range(3):
                    b.solve(); b.traceback(); b.sync()
                    ts.append(b.fill_ms())
                res = b.results()
                recs.append((res.copy(), b.transcripts(res)))
                out.append('%s %-34s fill %8.3f ms' % (title, b.kernel_name[:34], min(ts)))
        same = bool((recs[0][0] == recs[1][0]).all() and recs[0][1] == recs[1][1])
        print('one %d x %d LOCAL pair, transition / transversion matrix: %s | %s | records and transcripts equal: %s' % (n, len(m), out[0], out[1], same), flush=True)


if __name__ == '__main__':
    main()
