"""K1 on the config-2 batch: fill-kernel time of the packed 16-bit kernel per (diagonals per lane, pairs per wavefront)
choice (PWLIB_PACKED_BK=<bk>[s]), results compared with the default run.

    python tests/micro/k1_variants.py [choices ...]      e.g.  8 16s 20s 28s
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from biseqt_amd import synth, _pwlib as W          # noqa: E402
from biseqt_amd.batch import BatchAligner          # noqa: E402

o, m = synth.pair_batch(2, 10000, 2000)
kw = dict(alnmode=1, alntype=1, alphabet_len=4, diag_range=(-200, 200), match_score=1, mismatch_score=-3, go_score=-5,
          ge_score=-2, flags=W.PW_FLAG_PROFILE)
ref = None
for choice in (sys.argv[1:] or ['8', '16s', '20s', '28s']):
    os.environ['PWLIB_PACKED_BK'] = choice
    with BatchAligner(list(zip(o, m)), **kw) as b:
        ts = []
        for _ in range(6):
            b.solve(); b.traceback(); b.sync()
            ts.append(b.fill_ms())
        res = b.results()
        if ref is None:
            ref = res.copy()
        same = bool((res == ref).all())
        print('PWLIB_PACKED_BK=%-4s %-24s fill %.3f ms (min of 6: %.3f)  %.0f GCUPS  results %s'
              % (choice, b.kernel_name, float(np.median(ts)), min(ts), b.cells / min(ts) / 1e6, 'same' if same else 'DIFFER'), flush=True)
