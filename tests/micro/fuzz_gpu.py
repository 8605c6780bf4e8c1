"""Adversarial fuzz of the GPU path against the CPU oracle (test infrastructure; the oracle is the checker).

Batches of pairs with a shared problem description (mode, type, scores, kernel-forcing flags); pair shapes are
chosen to hit the places kernels break: alignments lying exactly on / next to a band edge, band widths that are
not multiples of the lane width, identical sequences, shifted copies, low-complexity repeats (many ties), empty
and one-letter sequences, clamped and infeasible bands, long pairs.

    python tests/micro/fuzz_gpu.py [seconds] [seed] [long | wide]      # long: banded pairs up to 14 kb, default kernels; wide: few pairs,
                                                                       # bands thousands of diagonals wide, all score sets and flags
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

SCORES = [(1, -3, -5, -2), (1, 0, 0, 0), (2, -1, 0, -1), (5, -4, -10, -1), (1, -1, -1, -1), (3, -2, 3, -4),
          (1, -1, 2, -3), (100, -100, -100, -100), (0, 0, 0, 0), (1, -3, 0, -2), (2, -3, -5, 0),
          (1, 6, -5, -2), (2, 3, -1, -1),      # mismatch > match: the API accepts it
          (-1, -2, -3, -1), (-1, 2, -2, -1), (-2, 0, -1, 0),   # ... and a negative match score
          (0.5, -0.25, -1.5, -0.75), (1.3862943611198906, -0.8754687373538999, -2.995732273553991, -0.6931471805599453),
          # round 3: dyadic score sets (scaled exactly onto the integer kernels), config 5's extension scores first
          (0.25, -1, 0, -1), (0.5, -1.5, -2.5, -1), (1.25, -0.75, -0.5, -0.25), (2, -3.125, -4.5, -0.0625), (0.75, 1.5, -0.25, -0.5)]


def make_matrix(rng, L):
    """Round 3: a substitution matrix instead of match / mismatch -- small integers (the packed kernels' byte rows when L <= 4),
    larger integers and larger alphabets (the wavefront kernels' table in LDS), dyadic and arbitrary floats."""
    kind = int(rng.integers(0, 5))
    if kind == 0:
        S = rng.integers(-6, 7, size=(L, L)).astype(np.float64)
    elif kind == 1:
        S = rng.integers(-40, 41, size=(L, L)).astype(np.float64)
    elif kind == 2:
        S = rng.integers(-3, 4, size=(L, L)).astype(np.float64); S = (S + S.T) / 2            # halves: dyadic
    elif kind == 3:
        S = rng.integers(-200, 300, size=(L, L)).astype(np.float64)
    else:
        S = np.round(rng.normal(0, 2, size=(L, L)), 3)
    if rng.random() < 0.7:
        S[np.arange(L), np.arange(L)] = np.abs(S[np.arange(L), np.arange(L)]) + 1
    return [[float(v) for v in row] for row in S]
FLAGS = [0, 0, 0, W.PW_FLAG_NO_PACKED16, W.PW_FLAG_FORCE_F64, W.PW_FLAG_FORCE_GENERIC, W.PW_FLAG_FORCE_TILED,
         W.PW_FLAG_FORCE_TILED | W.PW_FLAG_FORCE_F64, W.PW_FLAG_FORCE_STRIP, W.PW_FLAG_FORCE_STRIP]


def make_pair(rng, L, maxlen, wide=False):
    kind = rng.integers(0, 8)
    n = int(np.exp(rng.uniform(0, np.log(maxlen)))) if rng.random() < 0.9 else int(rng.integers(0, 3))
    if wide and rng.random() < 0.7:
        n = int(rng.integers(maxlen // 4, maxlen + 1))          # (the wide mode wants long sequences, not a log-uniform length)
    if kind == 0:                                   # unrelated
        o = rng.integers(0, L, n); m = rng.integers(0, L, max(0, n + int(rng.integers(-n // 2 - 1, n // 2 + 2))))
    elif kind == 1:                                 # identical
        o = rng.integers(0, L, n); m = o.copy()
    elif kind == 2:                                 # shifted copy: the alignment lies on diagonal +-k exactly
        o = rng.integers(0, L, n); k = int(rng.integers(0, min(n, 40) + 1))
        m = np.concatenate([rng.integers(0, L, k), o]) if rng.random() < 0.5 else o[k:].copy()
    elif kind == 3:                                 # low complexity: period-p repeat, many equal-score paths
        p = int(rng.integers(1, 4)); unit = rng.integers(0, L, p)
        o = np.resize(unit, n); m = np.resize(unit, max(0, n + int(rng.integers(-5, 6))))
    elif kind == 4:                                 # mutated copy
        o = rng.integers(0, L, n)
        m = synth.mutate(rng, o.astype(np.uint8), rng.uniform(0, .3), rng.uniform(0, .1), rng.uniform(0, .6), L=L) if n else o.copy()
    elif kind == 5:                                 # overlap: suffix of o = prefix of m
        o = rng.integers(0, L, n); k = int(rng.integers(0, n + 1))
        m = np.concatenate([o[k:], rng.integers(0, L, int(rng.integers(0, n + 1)))])
    elif kind == 6:                                 # containment
        m = rng.integers(0, L, n); a = int(rng.integers(0, n + 1))
        o = np.concatenate([rng.integers(0, L, int(rng.integers(0, 50))), m[a:], rng.integers(0, L, int(rng.integers(0, 50)))])
    else:                                           # one letter
        o = np.full(n, int(rng.integers(0, L))); m = np.full(max(0, n + int(rng.integers(-3, 4))), o[0] if n else 0)
    return np.asarray(o, np.uint8), np.asarray(m, np.uint8)


def make_band(rng, X, Y, o, m):
    r = rng.random()
    if r < 0.25:                                    # an edge right on / next to the main diagonal or X - Y
        c = [0, X - Y][int(rng.integers(0, 2))] + int(rng.integers(-2, 3))
        w = int(rng.integers(0, 70))
        lo, hi = (c - w, c) if rng.random() < 0.5 else (c, c + w)
    elif r < 0.5:
        c = int(rng.integers(-Y, X + 1)); w = int(rng.integers(0, 200))
        lo, hi = c - int(rng.integers(0, w + 1)), c + w
    elif r < 0.6:                                   # far outside: clamps / infeasible
        lo, hi = int(rng.integers(-Y - 50, X + 50)), int(rng.integers(-Y - 50, X + 50))
        lo, hi = min(lo, hi), max(lo, hi)
    else:
        lo, hi = -int(rng.integers(0, Y + 1)), int(rng.integers(0, X + 1))
    return lo, hi


def run(budget, seed, long_mode=False, max_batches=None):
    rng = np.random.default_rng(seed)
    t0 = time.time()
    nb = npairs = kernels = {}
    nb = npairs = nbad = 0
    last = t0
    while time.time() - t0 < budget and (max_batches is None or nb < max_batches):
        mode = int(rng.integers(0, 2))
        alntype = int(rng.integers(0, 7)) if mode == 0 else int(rng.integers(0, 3))
        sc = SCORES[int(rng.integers(0, len(SCORES)))]
        flags = FLAGS[int(rng.integers(0, len(FLAGS)))]
        L = [2, 4, 4, 20][int(rng.integers(0, 4))]
        maxlen = [40, 300, 1500, 4000][int(rng.integers(0, 4))] if mode == 1 else [40, 300, 1200][int(rng.integers(0, 3))]
        n = int(rng.integers(1, 25))
        if long_mode == 'wide':
            # few pairs, bands thousands of diagonals wide and far off the main diagonal, every score set: the multi-wavefront
            # kernels, strips and tiles on tables whose diagonals start and end at very different steps (round 3: the regime
            # of the fuzz's one late find)
            maxlen, n = [1500, 4000, 9000][int(rng.integers(0, 3))], int(rng.integers(1, 5))
            if mode == 0:
                maxlen = min(maxlen, 2500)
        elif long_mode:
            mode, maxlen, n = 1, 14000, int(rng.integers(1, 7))
            alntype = 1 if rng.random() < 0.7 else alntype % 3
            flags = 0 if rng.random() < 0.8 else flags
            sc = SCORES[int(rng.integers(0, 5))] if rng.random() < 0.8 else sc
        pairs, bands = [], []
        for _ in range(n):
            o, m = make_pair(rng, L, maxlen, long_mode == 'wide')
            pairs.append((o, m))
            bands.append(make_band(rng, len(o), len(m), o, m))
        subst = make_matrix(rng, L) if ((not long_mode or long_mode == 'wide') and rng.random() < 0.3) else None
        kw = dict(alnmode=mode, alntype=alntype, alphabet_len=L, go_score=sc[2], ge_score=sc[3], flags=flags, check_band=False)
        if subst is not None:
            kw['subst_scores'] = subst                         # (round 3: the strips take 4-letter byte-range matrices too; others fall through)
        else:
            kw.update(match_score=sc[0], mismatch_score=sc[1])
        if mode == 1:
            kw['diag_range'] = bands
        # kernel selection knobs the planner reads at batch creation: throughput kernels for small batches, the unscaled
        # packed kernel where the scaled one would be taken
        for name, p_on in (('PWLIB_LATENCY_MODE', 0.3), ('PWLIB_NO_SCALED16', 0.3), ('PWLIB_NO_DYADIC', 0.15), ('PWLIB_NO_PACKED_MAT', 0.15),
                           ('PWLIB_NO_PACKED_ANCHORED', 0.15)):
            os.environ.pop(name, None)
            if rng.random() < p_on:
                os.environ[name] = '0' if name == 'PWLIB_LATENCY_MODE' else '1'
        try:
            with BatchAligner(pairs, **kw) as b:
                kname = b.kernel_name
                kernels[kname.split('(')[0]] = kernels.get(kname.split('(')[0], 0) + 1
                res = b.run()
                txs = b.transcripts(res)
                rcs = [b.init_rc(k) for k in range(n)]
        except RuntimeError as e:
            print('batch failed:', e, dict(mode=mode, alntype=alntype, sc=sc, flags=flags)); nbad += 1
            continue
        nb += 1
        for k, (o, m) in enumerate(pairs):
            okw = dict(L=L, mode=mode, alntype=alntype, go=sc[2], ge=sc[3])
            if subst is not None:
                okw['subst'] = subst
            else:
                okw.update(match=sc[0], mismatch=sc[1])
            if mode == 1:
                okw['diag_range'] = bands[k]
            r = O.solve(o, m, **okw)
            npairs += 1
            why = None
            if r['init_rc'] != rcs[k]:
                why = 'init_rc %s != %s' % (rcs[k], r['init_rc'])
            elif r['init_rc'] == 0:
                if (int(res['opt_i'][k]), int(res['opt_j'][k])) != tuple(r['opt']):
                    why = 'opt (%d,%d) != %s' % (res['opt_i'][k], res['opt_j'][k], r['opt'])
                elif r['opt'][0] != -1:
                    if res['score'][k] != r['score']:
                        why = 'score %r != %r' % (res['score'][k], r['score'])
                    elif not r['would_panick'] and not r['tb_null'] and (
                            txs[k] != r['transcript'] or (int(res['origin_idx'][k]), int(res['mutant_idx'][k])) != (r['origin_idx'], r['mutant_idx'])):
                        why = 'transcript / start differ'
                    elif (r['would_panick'] or r['tb_null']) and txs[k] is not None and not r['would_panick']:
                        why = 'expected NULL traceback, got %r' % txs[k][:20]
            if why:
                nbad += 1
                print('MISMATCH %s | kernel %s mode %d type %d scores %s flags %d L %d band %s X %d Y %d matrix %s'
                      % (why, kname, mode, alntype, sc, flags, L, bands[k] if mode else None, len(o), len(m), subst))
                if nbad <= 3 and os.path.isdir(os.path.join(ROOT, 'gpurun_out')):
                    np.savez(os.path.join(ROOT, 'gpurun_out', 'fuzz_bad_%d.npz' % nbad), o=o, m=m, band=np.array(bands[k]),
                             meta=np.array([mode, alntype, flags, L]), sc=np.array(sc, dtype=np.float64))
        if time.time() - last > 30:
            last = time.time()
            print('... %d batches, %d pairs, %d bad' % (nb, npairs, nbad), flush=True)
    for name in ('PWLIB_LATENCY_MODE', 'PWLIB_NO_SCALED16', 'PWLIB_NO_DYADIC', 'PWLIB_NO_PACKED_MAT', 'PWLIB_NO_PACKED_ANCHORED'):      # (the knobs must not outlive the run: pytest calls it in-process)
        os.environ.pop(name, None)
    print('kernels: ' + ', '.join('%s x%d' % kv for kv in sorted(kernels.items(), key=lambda kv: -kv[1])))
    print('fuzz: %d batches, %d pairs, %d mismatches (seed %d, %.0f s)' % (nb, npairs, nbad, seed, time.time() - t0))
    return nb, npairs, nbad


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 12345
    mode = sys.argv[3] if len(sys.argv) > 3 else ''
    nb, npairs, nbad = run(budget, seed, 'wide' if mode == 'wide' else mode == 'long')
    sys.exit(1 if nbad else 0)


if __name__ == '__main__':
    main()
