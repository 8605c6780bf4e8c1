"""Word-Blot local similarity search at the config-5 shape (two 1 Mb sequences, 50 planted homologies of 2-20 kb at
80-95 % identity): seeds -> neighbourhood graph -> connected components -> segments, all device work timed;
the planted homologies must be recovered.

    python tests/micro/blot_bench.py [n] [k] [K_min]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from biseqt_amd import synth                        # noqa: E402
from biseqt_amd.blot import WordBlot                # noqa: E402
from biseqt_amd.pipeline import extend_segments     # noqa: E402
from biseqt_amd.sequence import Alphabet, Sequence  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    K_min = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
    rng = np.random.default_rng(5)
    s = rng.integers(0, 4, n).astype(np.uint8)
    t = rng.integers(0, 4, n).astype(np.uint8)
    planted = []
    for q in range(50):
        ln = int(rng.integers(2000, 20000))
        a = int(rng.integers(0, n - ln)); b = (q * (n // 50) + int(rng.integers(0, 1000))) % (n - ln)
        ident = rng.uniform(.8, .95)
        seg = synth.mutate(rng, s[a:a + ln], (1 - ident) * .7, (1 - ident) * .1, .3)
        seg = seg[:min(len(seg), n - b)]
        t[b:b + len(seg)] = seg
        planted.append((a, b, len(seg), ident))
    A = Alphabet('ACGT')
    t0 = time.perf_counter()
    S, T = Sequence(A, tuple(s.tolist())), Sequence(A, tuple(t.tolist()))
    t1 = time.perf_counter()
    wb = WordBlot(S, T, g_max=.1, sensitivity=.99, alphabet=A, wordlen=k)
    t2 = time.perf_counter()
    segs = list(wb.similar_segments(K_min, .7))
    t3 = time.perf_counter()
    segs2 = list(wb.similar_segments(K_min, .75))
    t4 = time.perf_counter()
    print('n=%d k=%d K_min=%d: %d seeds; Sequence objects %.2f s; index %.3f s (device build %.2f ms); similar_segments %.3f s '
          '(first: graph + components + scoring), %.3f s (graph cached); %d segments'
          % (n, k, K_min, wb.seed_count(), t1 - t0, t2 - t1, wb.build_ms(), t3 - t2, t4 - t3, len(segs)))
    hit = 0
    for (a, b, ln, ident) in planted:
        d = a - b
        mid = 2 * b + d + ln            # antidiagonal of the middle of the planted segment
        if any(sg['segment'][0][0] <= d <= sg['segment'][0][1] and sg['segment'][1][0] <= mid <= sg['segment'][1][1] for sg in segs):
            hit += 1
    print('planted homologies recovered: %d / %d; first segments:' % (hit, len(planted)))
    for sg in segs[:3]:
        print('  ', sg['segment'], 'p=%.3f' % sg['p'], 'z=(%.1f, %.1f)' % sg['scores'])
    # banded DP extension of every segment (experiments/blot_stats.py:438-470), one batch
    p_min = .7
    kw = dict(match_score=1. / p_min - 1, mismatch_score=-1, ge_score=-1, go_score=0)
    extend_segments(S, T, segs[:2], k, **kw)
    t5 = time.perf_counter()
    ext = extend_segments(S, T, segs, k, **kw)
    t6 = time.perf_counter()
    cells = sum((2 * r['diag_range'][1] + 1) * min(r['frame'][0][1] - r['frame'][0][0], r['frame'][1][1] - r['frame'][1][0]) for r in ext)
    ok = sum(1 for r in ext if r['truncated'] is not None)
    ident = [r['truncated'].transcript.count('M') / float(len(r['truncated'].transcript)) for r in ext if r['truncated'] is not None]
    print('banded extension of %d segments (f64 scores, B_GLOBAL): %.3f s wall incl. planning and D2H, ~%.2e cells; %d alignments, identity %.3f..%.3f'
          % (len(ext), t6 - t5, cells, ok, min(ident), max(ident)))
    wb.close()


if __name__ == '__main__':
    main()
