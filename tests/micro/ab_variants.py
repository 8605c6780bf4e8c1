"""Config 2's shape on the 32-bit, f64 and generic kernels (kernel-forcing flags), for A/B runs of two builds:
    PWLIB_SO=<library> python tests/micro/ab_variants.py [reps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from biseqt_amd import _pwlib as W            # noqa: E402
from biseqt_amd import synth                  # noqa: E402
from biseqt_amd.batch import BatchAligner     # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
origins, mutants = synth.pair_batch(2, 10000, 2000)
pairs = list(zip(origins, mutants))
base = dict(alnmode=1, alntype=1, alphabet_len=4, diag_range=(-200, 200), match_score=1, mismatch_score=-3, go_score=-5, ge_score=-2)
for title, flags in (('32-bit', W.PW_FLAG_NO_PACKED16), ('f64', W.PW_FLAG_FORCE_F64), ('generic', W.PW_FLAG_FORCE_GENERIC)):
    with BatchAligner(pairs, flags=flags | W.PW_FLAG_PROFILE, **base) as b:
        ts = []
        for _ in range(reps):
            b.solve(); b.sync()
            ts.append(b.fill_ms())
        print('%-8s %-40s fill %7.3f ms (best %7.3f) = %7.1f GCUPS' % (title, b.kernel_name, float(np.median(ts)), min(ts), b.cells / min(ts) / 1e6), flush=True)
