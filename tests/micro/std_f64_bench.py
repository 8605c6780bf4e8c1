"""Standard-mode tables on the 32-bit / f64 wavefront kernels (every block is a ramp block there): log-odds GLOBAL / LOCAL batches.
    python tests/micro/std_f64_bench.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from biseqt_amd import _pwlib as W, synth            # noqa: E402
from biseqt_amd.batch import BatchAligner            # noqa: E402

rng = synth.rng_for(11)
F64 = dict(match_score=1.2824, mismatch_score=-1.0361, go_score=-0.6931, ge_score=-1.2040)
BIG = dict(match_score=100, mismatch_score=-300, go_score=-500, ge_score=-200)
for title, n, length, kw in (('2000 x 250 GLOBAL f64', 2000, 250, dict(alnmode=0, alntype=0, **F64)),
                             ('2000 x 250 LOCAL f64', 2000, 250, dict(alnmode=0, alntype=1, **F64)),
                             ('1000 x 1000 GLOBAL f64', 1000, 1000, dict(alnmode=0, alntype=0, **F64)),
                             ('2000 x 250 GLOBAL int32 (big scores)', 2000, 250, dict(alnmode=0, alntype=0, **BIG)),
                             ('3000 x 400, band 61, B_GLOBAL f64', 3000, 400, dict(alnmode=1, alntype=0, diag_range=(-30, 30), **F64))):
    pairs = []
    for _ in range(n):
        o = synth.rand_seqs(rng, 1, length)[0]
        pairs.append((o, synth.mutate(rng, o, 0.05, 0.01, 0.2)))
    with BatchAligner(pairs, alphabet_len=4, check_band=False, flags=W.PW_FLAG_PROFILE, **kw) as b:
        ts = []
        for _ in range(4):
            b.solve(); b.traceback(); b.sync(); ts.append(b.fill_ms())
        print('%-40s %-40s fill %8.3f ms = %7.1f GCUPS' % (title, b.kernel_name, min(ts), b.cells / min(ts) / 1e6), flush=True)
