"""One config-2-shaped batch (10 000 pairs, 2 kb x ~2 kb, band radius 200, B_LOCAL, 1 / -3 / -5 / -2) solved a few times under
a forced kernel -- the program rocprofv3 is pointed at when a kernel other than the default one is profiled.

    python tests/micro/run_batch.py [f64|i32|generic|matrix|default] [reps] [pairs]
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from biseqt_amd import _pwlib as W            # noqa: E402
from biseqt_amd import synth                  # noqa: E402
from biseqt_amd.batch import BatchAligner     # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else 'default'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
origins, mutants = synth.pair_batch(2, n, 2000)
kw = dict(alnmode=1, alntype=1, alphabet_len=4, diag_range=(-200, 200), go_score=-5, ge_score=-2)
flags = {'f64': W.PW_FLAG_FORCE_F64, 'i32': W.PW_FLAG_NO_PACKED16, 'generic': W.PW_FLAG_FORCE_GENERIC}.get(which, 0)
if which == 'matrix':
    kw['subst_scores'] = [[1, -3, -2, -3], [-3, 1, -3, -2], [-2, -3, 1, -3], [-3, -2, -3, 1]]
else:
    kw.update(match_score=1, mismatch_score=-3)
with BatchAligner(list(zip(origins, mutants)), flags=flags | W.PW_FLAG_PROFILE, **kw) as b:
    for _ in range(reps):
        b.solve(); b.traceback(); b.sync()
        print('%s: fill %.3f ms, traceback %.3f ms, %s' % (which, b.fill_ms(), b.trace_ms(), b.kernel_name), flush=True)
