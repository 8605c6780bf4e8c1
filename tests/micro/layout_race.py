"""The planner's lane layout of the packed kernels against every layout that can be forced (PWLIB_PACKED_BK=<bk>[s]: <bk>
diagonals per lane, one pair per wavefront or lane-packed), on batches of many banded pairs -- one band width and mixed widths.

    python tests/micro/layout_race.py [reps]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from biseqt_amd import _pwlib as W            # noqa: E402
from biseqt_amd import synth                  # noqa: E402
from biseqt_amd.batch import BatchAligner     # noqa: E402


def run(pairs, bands, alntype, forced, reps):
    os.environ.pop('PWLIB_PACKED_BK', None)
    if forced:
        os.environ['PWLIB_PACKED_BK'] = forced
    try:
        with BatchAligner(pairs, flags=W.PW_FLAG_PROFILE, alnmode=1, alntype=alntype, diag_range=bands, alphabet_len=4,
                          match_score=1, mismatch_score=-3, go_score=-5, ge_score=-2, check_band=False) as b:
            if forced and 'k_fill16' not in b.kernel_name:
                return None
            ts = []
            for _ in range(reps):
                b.solve(); b.traceback(); b.sync()
                ts.append(b.fill_ms() + b.trace_ms())
            return min(ts), b.kernel_name
    except RuntimeError:
        return None
    finally:
        os.environ.pop('PWLIB_PACKED_BK', None)


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    rng = np.random.default_rng(31)
    cases = [('20000 x 5 kb, bands 9..111 (mixed), B_OVERLAP', 20000, 5000, (4, 55), 2),
             ('50000 x 5 kb, bands 9..111 (mixed), B_OVERLAP', 50000, 5000, (4, 55), 2),
             ('5000 x 2 kb, radius 20..300 (mixed), B_LOCAL', 5000, 2000, (20, 300), 1),
             ('3000 x 1 kb, radius 120, B_LOCAL', 3000, 1000, (120, 120), 1),
             ('50000 x 500, radius 30, B_LOCAL', 50000, 500, (30, 30), 1),
             ('2000 x 3 kb, radius 400, B_LOCAL', 2000, 3000, (400, 400), 1),
             ('8000 x 2 kb, radius 60, B_OVERLAP', 8000, 2000, (60, 60), 2),
             ('1500 x 2 kb, radius 50..600 (mixed), B_LOCAL', 1500, 2000, (50, 600), 1)]
    worst = 0.0
    for title, n, length, (rlo, rhi), alntype in cases:
        origins, mutants = synth.pair_batch(7, n, length)
        pairs = list(zip(origins, mutants))
        radii = rng.integers(rlo, rhi + 1, n)
        bands = [(-int(r), int(r)) for r in radii]
        base = run(pairs, bands, alntype, None, reps)
        alts = {}
        for bk in (4, 8, 12, 16, 20, 24, 28, 32):
            for suffix in ('', 's'):
                r = run(pairs, bands, alntype, '%d%s' % (bk, suffix), reps)
                if r is not None:
                    alts['%d%s' % (bk, suffix)] = r
        best = min(alts.items(), key=lambda kv: kv[1][0])
        loss = base[0] / best[1][0] - 1.0
        worst = max(worst, loss)
        print('%-48s planner %7.3f ms (%s) | best forced %s: %7.3f ms (%s) | planner %+5.1f %% | %s'
              % (title, base[0], base[1], best[0], best[1][0], best[1][1], 100 * loss,
                 '  '.join('%s %.3f' % (k, v[0]) for k, v in sorted(alts.items(), key=lambda kv: (int(kv[0].rstrip('s')), kv[0])))), flush=True)
    print('worst planner pick: %.1f %% behind the best forced layout' % (100 * worst))


if __name__ == '__main__':
    main()
