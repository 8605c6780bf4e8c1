"""Sweep of problem shapes through the planner: kernel chosen, fill + traceback time, ns per cell -- to spot shapes the planner
serves badly (a neighbouring shape that is much cheaper per cell).

    python tests/micro/shape_sweep.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from biseqt_amd import synth, _pwlib as W          # noqa: E402
from biseqt_amd.batch import BatchAligner          # noqa: E402

rng = synth.rng_for(79)


def run(tag, pairs, **kw):
    with BatchAligner(pairs, flags=W.PW_FLAG_PROFILE, **kw) as b:
        ts = []
        for _ in range(3):
            b.solve(); b.traceback(); b.sync()
            ts.append(b.fill_ms() + b.trace_ms())
        t = min(ts)
        print('%-44s %-30s %9.3f ms  %7.3f ns/cell' % (tag, b.kernel_name[:30], t, t * 1e6 / max(b.cells, 1)), flush=True)


def mk(n, count):
    out = []
    for _ in range(count):
        o = synth.rand_seqs(rng, 1, n)[0]
        out.append((o, synth.mutate(rng, o, 0.05, 0.02, 0.4)))
    return out


base = dict(alphabet_len=4, match_score=1, mismatch_score=-3, go_score=-5, ge_score=-2)
for count in (1, 16, 300):
    for n in (2000, 10000, 30000):
        if n * count > 3500000:
            continue
        P = mk(n, count)
        for r in (50, 200, 600, 1500, 4000):
            if r * 2 > n:
                continue
            for name, t in (('B_LOCAL', 1), ('B_GLOBAL', 0), ('B_OVERLAP', 2)):
                run('%3d x %5d banded r=%4d %s' % (count, n, r, name), P, alnmode=1, alntype=t, diag_range=(-r, r), **base)
        if n <= 10000 and count <= 16:
            run('%3d x %5d standard LOCAL' % (count, n), P, alnmode=0, alntype=1, **base)
            run('%3d x %5d standard GLOBAL' % (count, n), P, alnmode=0, alntype=0, **base)
            run('%3d x %5d standard LOCAL f64 scores' % (count, n), P, alnmode=0, alntype=1, alphabet_len=4, match_score=0.5,
                mismatch_score=-1.25, go_score=-2, ge_score=-1)
