"""Pin oracle/pw_oracle.c against the compiled reference (oracle/_ref/pwlib_ref.so) -- TEST INFRA.

    python -m oracle.check_vs_ref [n_problems] [seed]

Random problems over all 7 standard + 3 banded alignment types, go in {<0, 0, >0}, sub-frames,
clamped and infeasible bands.  Compares init rc, clamped band, optimal cell, score, transcript,
start indices and (STD mode) the whole score table.  Inputs on which the reference would exit(1)
(pw.c:132-134) are predicted by the oracle (`would_panick`) and only solved, not traced back.
"""
import sys

import numpy as np

from . import oracle as O
from . import ref_driver as R


def random_problem(rng, maxlen=24):
    L = int(rng.choice([1, 2, 4]))
    n, m = int(rng.integers(0, maxlen + 1)), int(rng.integers(0, maxlen + 1))
    origin = rng.integers(0, L, n).tolist()
    mutant = rng.integers(0, L, m).tolist()
    if n and m and rng.random() < 0.5:     # plant similarity
        k = int(rng.integers(1, min(n, m) + 1))
        i, j = int(rng.integers(0, n - k + 1)), int(rng.integers(0, m - k + 1))
        mutant[j:j + k] = origin[i:i + k]
    kw = dict(L=L)
    if rng.random() < 0.25:
        kw['subst'] = rng.integers(-4, 5, (L, L)).astype(float).tolist()
    else:
        kw['match'] = float(rng.choice([1, 2, 5]))
        kw['mismatch'] = float(rng.choice([0, -1, -3]))
    kw['go'] = float(rng.choice([0, 0, -1, -4, -5, 2]))
    kw['ge'] = float(rng.choice([0, -1, -2, 1]))
    if rng.random() < 0.3 and n and m:
        a, b = sorted(rng.integers(0, n + 1, 2).tolist())
        kw['origin_range'] = (a, b)
        a, b = sorted(rng.integers(0, m + 1, 2).tolist())
        kw['mutant_range'] = (a, b)
    if rng.random() < 0.5:
        kw['mode'] = R.STD_MODE
        kw['alntype'] = int(rng.integers(0, 7))
    else:
        kw['mode'] = R.BANDED_MODE
        kw['alntype'] = int(rng.integers(0, 3))
        lo, hi = sorted(rng.integers(-m - 3, n + 4, 2).tolist())
        kw['diag_range'] = (lo, hi)
    return origin, mutant, kw


def compare(origin, mutant, kw, reflib):
    o = O.solve(origin, mutant, want_table=(kw['mode'] == R.STD_MODE), **kw)
    P = R.Problem(origin, mutant, **kw)
    panick = bool(o.get('would_panick'))
    r = R.run(reflib, P, want_table=True, do_traceback=not panick)
    errs = []
    if o.get('maskrule_ok') is False:
        errs.append(('maskrule', False, True))
    for key in ('init_rc', 'opt', 'score', 'band', 'num_rows'):
        if o.get(key) != r.get(key):
            errs.append((key, o.get(key), r.get(key)))
    if not panick and r['init_rc'] == 0 and r['opt'] is not None and r['opt'][0] != -1:
        for key in ('tb_null', 'transcript', 'origin_idx', 'mutant_idx'):
            if o.get(key) != r.get(key):
                errs.append((key, o.get(key), r.get(key)))
    if 'table' in r and o.get('H') is not None:
        Y = len(r['table'][0]) - 1
        H = o['H'].reshape(-1, Y + 1)
        for i, row in enumerate(r['table']):
            for j, v in enumerate(row):
                hv = H[i, j]
                if (v is None) != bool(np.isnan(hv)) or (v is not None and v != hv):
                    errs.append(('table', (i, j), hv, v))
                    break
    return errs, panick


def main(n=3000, seed=0):
    rng = np.random.default_rng(seed)
    reflib = R.load()
    bad = npanick = 0
    for t in range(n):
        origin, mutant, kw = random_problem(rng)
        errs, panick = compare(origin, mutant, kw, reflib)
        npanick += panick
        if errs:
            bad += 1
            if bad <= 5:
                print('MISMATCH', origin, mutant, kw, errs[:3])
    print('problems=%d mismatches=%d would_panick=%d' % (n, bad, npanick))
    return bad


if __name__ == '__main__':
    a = [int(v) for v in sys.argv[1:]]
    sys.exit(1 if main(*a) else 0)
