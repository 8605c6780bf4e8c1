"""One standard-mode pair of medium size: the default kernel choice against the strip pipeline forced.

    python tests/micro/mid_pairs.py [sizes ...]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from biseqt_amd import synth, _pwlib as W          # noqa: E402
from biseqt_amd.batch import BatchAligner          # noqa: E402

rng = synth.rng_for(77)
for n in [int(a) for a in sys.argv[1:]] or [500, 1000, 2000, 3000, 5000, 8000]:
    o = synth.rand_seqs(rng, 1, n)[0]
    m = synth.mutate(rng, o, 0.07, 0.02, 0.4)
    out = []
    ref = None
    for name, flags in (('default', 0), ('strips', W.PW_FLAG_FORCE_STRIP)):
        with BatchAligner([(o, m)], alnmode=0, alntype=1, alphabet_len=4, match_score=1, mismatch_score=-3, go_score=-5, ge_score=-2,
                          flags=flags | W.PW_FLAG_PROFILE) as b:
            ts, tt = [], []
            for _ in range(4):
                b.solve(); b.traceback(); b.sync()
                ts.append(b.fill_ms()); tt.append(b.trace_ms())
            res = b.results()
            tx = b.transcripts(res)[0]
            if ref is None:
                ref = (res.copy(), tx)
            same = bool((res == ref[0]).all()) and tx == ref[1]
            out.append('%s: %-34s fill %7.3f ms trace %6.3f ms %s' % (name, b.kernel_name[:34], min(ts), min(tt), '' if same else 'DIFFERS'))
    print('n = %5d   %s   |   %s' % (n, out[0], out[1]), flush=True)
