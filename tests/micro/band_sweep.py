"""Diagnostic (not a test): fill-kernel GCUPS for several band radii, with and without lane packing.
    python tests/micro/band_sweep.py            (on a GPU box)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from biseqt_amd import synth, _pwlib as W
from biseqt_amd.batch import BatchAligner
radius, n = int(sys.argv[1]), int(sys.argv[2])
origins, mutants = synth.pair_batch(5, n, 2000)
b = BatchAligner(list(zip(origins, mutants)), alnmode=1, alntype=1, alphabet_len=4, diag_range=(-radius, radius),
                 match_score=1, mismatch_score=-3, go_score=-5, ge_score=-2, flags=W.PW_FLAG_PROFILE)
for _ in range(3):
    b.solve(); b.traceback(); b.sync()
ms = []
for _ in range(5):
    b.solve(); b.traceback(); b.sync(); ms.append((b.fill_ms(), b.trace_ms()))
f = np.mean([m[0] for m in ms]); t = np.mean([m[1] for m in ms])
print("radius %%4d  %%-24s fill %%7.3f ms  %%8.1f GCUPS   trace %%6.3f ms" %% (radius, b.kernel_name, f, b.cells / f / 1e6, t))
''' % ROOT

for radius in (16, 50, 100, 200, 400):
    for env in ({}, {'PWLIB_PACKED_BK': os.environ.get('SWEEP_FORCE', '8')}):
        e = dict(os.environ, **env)
        if not env:
            e.pop('PWLIB_PACKED_BK', None)
        r = subprocess.run([sys.executable, '-c', CHILD, str(radius), '10000'], env=e, stdout=subprocess.PIPE,
                           stderr=subprocess.DEVNULL, universal_newlines=True)
        print(('auto   ' if not env else 'BK=%s   ' % env['PWLIB_PACKED_BK']) + r.stdout.strip())
