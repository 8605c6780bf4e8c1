"""Second sweep: many small pairs (the planner's throughput side).  python tests/micro/shape_sweep_small.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from biseqt_amd import synth, _pwlib as W          # noqa: E402
from biseqt_amd.batch import BatchAligner          # noqa: E402

rng = synth.rng_for(80)
base = dict(alphabet_len=4, match_score=1, mismatch_score=-3, go_score=-5, ge_score=-2)
for n in (100, 300, 1000):
    for count in (1000, 20000):
        if n * n * count > 3e10:
            continue
        o = synth.rand_seqs(rng, count, n)
        P = [(a, synth.mutate(rng, a, 0.05, 0.02, 0.4)) for a in o]
        shapes = [('standard LOCAL', dict(alnmode=0, alntype=1)), ('standard GLOBAL', dict(alnmode=0, alntype=0)),
                  ('standard OVERLAP', dict(alnmode=0, alntype=4))]
        for r in (10, 40):
            shapes.append(('banded r=%d B_LOCAL' % r, dict(alnmode=1, alntype=1, diag_range=(-r, r))))
            shapes.append(('banded r=%d B_OVERLAP' % r, dict(alnmode=1, alntype=2, diag_range=(-r, r))))
        for name, kw in shapes:
            kw = dict(kw); kw.update(base)
            with BatchAligner(P, flags=W.PW_FLAG_PROFILE, **kw) as b:
                ts = []
                for _ in range(3):
                    b.solve(); b.traceback(); b.sync()
                    ts.append((b.fill_ms(), b.trace_ms()))
                f, t = min(ts)
                print('%6d x %5d %-24s %-30s fill %8.3f ms trace %7.3f ms  %8.1f GCUPS' % (count, n, name, b.kernel_name[:30], f, t, b.cells / (f + t) / 1e6), flush=True)
