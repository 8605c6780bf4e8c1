"""Batches of a few standard-mode pairs of medium size: the workgroup-per-pair kernels (all pairs at once; 32-bit and 16-bit body) against the strip
pipeline (pairs one after another, each with the whole chip).  Data behind the planner's rule (pwlib_api.cpp, `few`).

    python tests/micro/few_pairs.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from biseqt_amd import synth, _pwlib as W          # noqa: E402
from biseqt_amd.batch import BatchAligner          # noqa: E402

rng = synth.rng_for(78)
kw = dict(alnmode=0, alntype=1, alphabet_len=4, match_score=1, mismatch_score=-3, go_score=-5, ge_score=-2)
for n in (300, 500, 1000, 1200, 2000, 4000):
    for count in (1, 4, 8, 16, 32, 64):
        if n * count > 70000:
            continue
        pairs = []
        for _ in range(count):
            o = synth.rand_seqs(rng, 1, n)[0]
            pairs.append((o, synth.mutate(rng, o, 0.07, 0.02, 0.4)))
        out = []
        for name, flags, env in (('workgroups', 0, '1'), ('packed-mw', 0, 'p'), ('strips', W.PW_FLAG_FORCE_STRIP, ''), ('planner', 0, '')):
            os.environ.pop('PWLIB_NO_SMALL_STRIP', None)
            os.environ.pop('PWLIB_NO_PACKED_MW', None)
            if env == '1':
                os.environ['PWLIB_NO_SMALL_STRIP'] = '1'; os.environ['PWLIB_NO_PACKED_MW'] = '1'
            elif env == 'p':
                os.environ['PWLIB_NO_SMALL_STRIP'] = '1'
            with BatchAligner(pairs, flags=flags | W.PW_FLAG_PROFILE, **kw) as b:
                ts = []
                for _ in range(3):
                    b.solve(); b.traceback(); b.sync()
                    ts.append(b.fill_ms() + b.trace_ms())
                out.append('%s %8.3f ms (%s)' % (name, min(ts), b.kernel_name[:22]))
        for e in ('PWLIB_NO_SMALL_STRIP', 'PWLIB_NO_PACKED_MW'):
            os.environ.pop(e, None)
        print('n = %5d x %2d pairs   %s' % (n, count, '   '.join(out)), flush=True)
