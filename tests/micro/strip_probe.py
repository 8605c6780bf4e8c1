"""Timing probe of the strip pipeline (K2c): shapes that separate the time per step of one strip (a single strip over
many columns) from the lag a FIFO hop adds (many strips over few columns).

    python tests/micro/strip_probe.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from biseqt_amd import synth, _pwlib as W          # noqa: E402
from biseqt_amd.batch import BatchAligner          # noqa: E402

rng = synth.rng_for(33)
shapes = [(63, 100000), (127, 100000), (639, 100000), (6399, 100000), (20000, 2000), (60000, 2000), (60000, 200), (9000, 9000), (30000, 30000)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]]
for X, Y in shapes:
    o = synth.rand_seqs(rng, 1, X)[0]
    m = synth.rand_seqs(rng, 1, Y)[0]
    with BatchAligner([(o, m)], alnmode=0, alntype=1, alphabet_len=4, match_score=1, mismatch_score=-3, go_score=-5, ge_score=-2,
                      flags=W.PW_FLAG_PROFILE | W.PW_FLAG_FORCE_STRIP) as b:
        ts = []
        for _ in range(3):
            b.solve(); b.sync()
            ts.append(b.fill_ms())
        nstrips = (X + 1 + 63) // 64
        t = min(ts)
        print('X %6d Y %6d strips %5d  fill %8.3f ms  per column-step of one strip %6.1f ns  (cells %.3g, %.1f GCUPS)'
              % (X, Y, nstrips, t, t * 1e6 / (Y + 64), b.cells, b.cells / t / 1e6), flush=True)
