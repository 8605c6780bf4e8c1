"""The strip pipeline (K2c, pw_strip.h) on the GPU against the CPU oracle: standard-mode problems of every alignment
type forced through it (PW_FLAG_FORCE_STRIP), one pair per batch and several pairs per batch (they run one after
another), plus re-runs of the same batch (the FIFO tags of an earlier solve must never be taken for fresh ones).

    python tests/micro/strip_check.py [cases] [seed] [maxlen]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from biseqt_amd import synth, _pwlib as W          # noqa: E402
from biseqt_amd.batch import BatchAligner          # noqa: E402
from oracle import oracle as O                     # noqa: E402

SCORES = [(1, -3, -5, -2), (1, 0, 0, 0), (2, -1, 0, -1), (5, -4, -10, -1), (1, -1, -1, -1), (1, -3, 0, -2), (1, 6, -5, -2)]


def random_matrix(rng, L):
    """A substitution matrix the strip kernel's byte rows take: integers within a signed byte (round 3)."""
    hi = int(rng.choice([3, 9, 60, 127]))
    S = rng.integers(-hi, hi + 1, size=(L, L))
    if rng.random() < 0.7:
        S[np.arange(L), np.arange(L)] = rng.integers(1, hi + 1, size=L)
    if rng.random() < 0.3:
        S = (S + S.T) // 2
    return [[float(v) for v in row] for row in S]


def run(cases=60, seed=1, maxlen=700, matrices=False):
    rng = np.random.default_rng(seed)
    nbad = npairs = 0
    t0 = time.time()
    for c in range(cases):
        alntype = c % 7
        sc = SCORES[int(rng.integers(0, len(SCORES)))]
        n = int(rng.integers(1, 4))
        pairs = []
        for _ in range(n):
            X = int(rng.integers(0, maxlen)) if rng.random() < 0.9 else int(rng.integers(0, 4))
            o = rng.integers(0, 4, X).astype(np.uint8)
            kind = int(rng.integers(0, 4))
            if kind == 0:
                m = synth.mutate(rng, o, 0.1, 0.05, 0.3) if X else rng.integers(0, 4, int(rng.integers(0, 9))).astype(np.uint8)
            elif kind == 1:
                k = int(rng.integers(0, X + 1))
                m = np.concatenate([o[k:], rng.integers(0, 4, int(rng.integers(0, 90))).astype(np.uint8)])
            elif kind == 2:
                m = rng.integers(0, 4, int(rng.integers(0, maxlen))).astype(np.uint8)
            else:
                m = o.copy()
            pairs.append((o, m))
        kw = dict(alnmode=0, alntype=alntype, alphabet_len=4, match_score=sc[0], mismatch_score=sc[1], go_score=sc[2],
                  ge_score=sc[3], flags=W.PW_FLAG_FORCE_STRIP)
        okw = dict(match=sc[0], mismatch=sc[1])
        if matrices:
            S = random_matrix(rng, 4)
            kw = dict(alnmode=0, alntype=alntype, alphabet_len=4, subst_scores=S, go_score=min(sc[2], 0), ge_score=sc[3],
                      flags=W.PW_FLAG_FORCE_STRIP)
            okw = dict(subst=S)
            sc = (None, None, min(sc[2], 0), sc[3])
        with BatchAligner(pairs, **kw) as b:
            assert 'k_fill_strip' in b.kernel_name, b.kernel_name
            for rep in range(2):                      # the second run re-uses the FIFO rows of the first
                res = b.run()
                txs = b.transcripts(res)
                for k, (o, m) in enumerate(pairs):
                    r = O.solve(o, m, L=4, mode=0, alntype=alntype, go=sc[2], ge=sc[3], **okw)
                    npairs += 1
                    why = None
                    if (int(res['opt_i'][k]), int(res['opt_j'][k])) != tuple(r['opt']):
                        why = 'opt (%d,%d) != %s' % (res['opt_i'][k], res['opt_j'][k], r['opt'])
                    elif r['opt'][0] != -1:
                        if res['score'][k] != r['score']:
                            why = 'score %r != %r' % (res['score'][k], r['score'])
                        elif not r['would_panick'] and not r['tb_null'] and (
                                txs[k] != r['transcript'] or (int(res['origin_idx'][k]), int(res['mutant_idx'][k])) != (r['origin_idx'], r['mutant_idx'])):
                            why = 'transcript / start differ'
                        elif r['tb_null'] and txs[k] is not None:
                            why = 'expected no transcript'
                    if why:
                        nbad += 1
                        print('MISMATCH', why, 'type', alntype, 'scores', sc, 'X', len(o), 'Y', len(m), 'rep', rep, flush=True)
    print('strip check: %d cases, %d pair-runs, %d mismatches, %.1f s' % (cases, npairs, nbad, time.time() - t0))
    return nbad


if __name__ == '__main__':
    a = sys.argv[1:]
    sys.exit(1 if run(int(a[0]) if a else 60, int(a[1]) if len(a) > 1 else 1, int(a[2]) if len(a) > 2 else 700) else 0)
