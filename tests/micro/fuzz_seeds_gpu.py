"""Adversarial fuzz of the seeds / band selection / local similarity / batched overlap GPU paths against their
oracles (test infrastructure).  python tests/micro/fuzz_seeds_gpu.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from biseqt_amd.blot import WordBlot, WordBlotOverlap     # noqa: E402
from biseqt_amd.overlap import overlap_bands              # noqa: E402
from biseqt_amd.seeds import SeedIndex                    # noqa: E402
from biseqt_amd.sequence import Alphabet, Sequence        # noqa: E402
from biseqt_amd import synth                              # noqa: E402
from oracle import blot_oracle as BO, seeds_oracle as SO  # noqa: E402


def make(rng, L, maxlen):
    n = int(np.exp(rng.uniform(np.log(4), np.log(maxlen))))
    kind = int(rng.integers(0, 6))
    s = rng.integers(0, L, n)
    if kind == 0:
        t = rng.integers(0, L, max(1, n + int(rng.integers(-n // 2, n // 2 + 1))))
    elif kind == 1:                                   # shared block at random offsets
        ln = int(rng.integers(1, n + 1)); a = int(rng.integers(0, n - ln + 1))
        t = np.concatenate([rng.integers(0, L, int(rng.integers(0, n))), synth.mutate(rng, s[a:a + ln].astype(np.uint8), rng.uniform(0, .15), rng.uniform(0, .05), .3, L=L), rng.integers(0, L, int(rng.integers(0, n)))])
    elif kind == 2:                                   # suffix-prefix overlap
        k = int(rng.integers(0, n)); t = np.concatenate([s[k:], rng.integers(0, L, int(rng.integers(0, n + 1)))])
    elif kind == 3:                                   # tandem repeats
        u = rng.integers(0, L, int(rng.integers(1, 6))); s = np.resize(u, n); t = np.resize(np.roll(u, int(rng.integers(0, 3))), max(1, n + int(rng.integers(-9, 10))))
    elif kind == 4:                                   # two letters only
        s = s % 2; t = rng.integers(0, 2, max(1, n + int(rng.integers(-20, 21))))
    else:                                             # two homologies on different diagonals
        t = np.concatenate([s[n // 2:], rng.integers(0, L, 30), s[:n // 2]])
    if len(t) == len(s) and (t == s).all():
        t = np.concatenate([t, [int(t[-1] + 1) % L]])
    return np.asarray(s, np.uint8), np.asarray(t, np.uint8)


def run(budget, seed):
    rng = np.random.default_rng(seed)
    alphs = [Alphabet('ACGT'), Alphabet('AB'), Alphabet([chr(97 + i) for i in range(12)])]
    t0 = time.time(); n = bad = 0
    while time.time() - t0 < budget:
        A = alphs[int(rng.integers(0, 3))]; L = len(A)
        k = int(rng.integers(2, 10)) if L <= 4 else int(rng.integers(1, 4))
        s, t = make(rng, L, [60, 400, 1500][int(rng.integers(0, 3))])
        g_max, sens = float(rng.choice([.05, .1, .2, .3])), float(rng.choice([.9, .99, .999]))
        S, T = Sequence(A, tuple(s.tolist())), Sequence(A, tuple(t.tolist()))
        why = None
        rows, sc = SO.seed_rows(s.tolist(), t.tolist(), k, L)
        if len(rows) > 300000:
            continue
        idx = SeedIndex(S, T, wordlen=k, alphabet=A)
        if [tuple(r) for r in idx.rows().tolist()] != rows:
            why = 'rows differ'
        d0 = int(rng.integers(-len(t), len(s) + 1)); a0 = int(rng.integers(0, len(s) + len(t)))
        db, ab = (d0, d0 + int(rng.integers(0, 40))), (a0, a0 + int(rng.integers(0, 300)))
        if why is None and idx.seed_count(d_band=db, a_band=ab) != SO.seed_count(rows, db, ab):
            why = 'band count differs'
        idx.close()
        if why is None and len(rows) <= 20000:
            wb = WordBlotOverlap(S, T, g_max=g_max, sensitivity=sens, alphabet=A, wordlen=k)
            got = wb.highest_scoring_overlap_band(); wb.close()
            exp = BO.highest_scoring_overlap_band(s.tolist(), t.tolist(), k, L, g_max, sens)
            if (got is None) != (exp is None) or (got is not None and (got['d_band'] != exp['d_band'] or got['p'] != exp['p'] or got['score'] != exp['score'])):
                why = 'overlap band differs: %r vs %r' % (got, exp)
            if why is None:
                gb = overlap_bands([s, t], [(0, 1), (1, 0)], k, A, g_max, sens)[0]
                if (gb is None) != (exp is None) or (gb is not None and (gb['d_band'] != exp['d_band'] or gb['p'] != exp['p'] or gb['score'] != exp['score'])):
                    why = 'batched overlap band differs: %r vs %r' % (gb, exp)
        if why is None and len(rows) <= 6000:
            K = int(rng.integers(5, 200)); p_min = float(rng.choice([.3, .6, .8, .95]))
            wb = WordBlot(S, T, g_max=g_max, sensitivity=sens, alphabet=A, wordlen=k)
            gs = list(wb.similar_segments(K, p_min)); wb.close()
            es = BO.similar_segments(s.tolist(), t.tolist(), k, L, g_max, sens, K, p_min)
            if [g['segment'] for g in gs] != [e['segment'] for e in es]:
                why = 'segments differ (K=%d p_min=%g)' % (K, p_min)
            elif any(abs(g['p'] - e['p']) > 1e-12 * max(abs(e['p']), 1e-300) for g, e in zip(gs, es)):
                why = 'segment p differs'
        n += 1
        if why:
            bad += 1
            print('MISMATCH %s | L=%d k=%d |S|=%d |T|=%d g=%g sens=%g' % (why[:300], L, k, len(s), len(t), g_max, sens), flush=True)
            if bad <= 3 and os.path.isdir(os.path.join(ROOT, 'gpurun_out')):
                np.savez(os.path.join(ROOT, 'gpurun_out', 'fuzz_seeds_bad_%d.npz' % bad), s=s, t=t, meta=np.array([L, k]), gs=np.array([g_max, sens]))
    print('fuzz seeds: %d cases, %d mismatches (seed %d)' % (n, bad, seed))
    return n, bad


if __name__ == '__main__':
    n, bad = run(float(sys.argv[1]) if len(sys.argv) > 1 else 60., int(sys.argv[2]) if len(sys.argv) > 2 else 7)
    sys.exit(1 if bad else 0)
