"""K2a with the 16-bit body (k_fill16_mw: several wavefronts per pair): batches of standard-mode / wide-band pairs against the
32-bit kernels (PW_FLAG_NO_PACKED16) record for record, a sample against the oracle, and the timing of both.

    python tests/micro/packed_mw_check.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from biseqt_amd import synth, _pwlib as W          # noqa: E402
from biseqt_amd.batch import BatchAligner          # noqa: E402
from oracle import oracle as O                     # noqa: E402

rng = synth.rng_for(91)
bad = 0
for count, n, mode, alntype, band, sc in ((400, 1500, 0, 1, None, (1, -3, -5, -2)), (400, 1200, 0, 0, None, (2, -3, -4, -1)),
                                          (400, 1800, 0, 4, None, (1, -1, -1, -1)), (300, 6000, 1, 1, (-1500, 1400), (1, -3, -5, -2)),
                                          (300, 5000, 1, 2, (-1200, 1300), (1, -3, 0, -2)), (300, 3000, 1, 0, (-2500, 2500), (1, -2, -3, -1)),
                                          (1000, 2000, 0, 1, None, (1, -3, -5, -2)), (300, 4000, 0, 1, None, (1, -3, -5, -2))):
    pairs = []
    for k in range(count):
        m_len = n if k % 5 else int(rng.integers(1, n))
        o = synth.rand_seqs(rng, 1, n if k % 7 else int(rng.integers(0, n)))[0]
        if k % 3 == 0:
            m = synth.rand_seqs(rng, 1, m_len)[0]
        else:
            m = synth.mutate(rng, o, 0.05, 0.02, 0.4)
        pairs.append((o, m))
    kw = dict(alnmode=mode, alntype=alntype, alphabet_len=4, match_score=sc[0], mismatch_score=sc[1], go_score=sc[2], ge_score=sc[3],
              check_band=False)
    if band is not None:
        kw['diag_range'] = band
    runs = []
    for flags in (0, W.PW_FLAG_NO_PACKED16):
        with BatchAligner(pairs, flags=flags | W.PW_FLAG_PROFILE, **kw) as b:
            name = b.kernel_name
            ts = []
            for _ in range(2):
                b.solve(); b.traceback(); b.sync(); ts.append(b.fill_ms())
            res = b.results()
            runs.append((name, res.copy(), b.transcripts(res), min(ts)))
    same = bool((runs[0][1] == runs[1][1]).all()) and runs[0][2] == runs[1][2]
    nbad = 0
    for k in range(0, count, 41):
        okw = dict(L=4, mode=mode, alntype=alntype, match=sc[0], mismatch=sc[1], go=sc[2], ge=sc[3])
        if band is not None:
            okw['diag_range'] = band
        r = O.solve(pairs[k][0], pairs[k][1], **okw)
        g = runs[0][1]
        if r['init_rc'] != 0:
            continue
        if (int(g['opt_i'][k]), int(g['opt_j'][k])) != tuple(r['opt']) or (r['opt'][0] >= 0 and g['score'][k] != r['score']):
            nbad += 1
        elif r['opt'][0] >= 0 and not r['would_panick'] and (runs[0][2][k] or '') != (r['transcript'] or ''):
            nbad += 1
    bad += (0 if same else 1) + nbad
    print('%4d x %5d mode %d type %d band %-14s %-34s %8.2f ms | %-34s %8.2f ms  %s oracle-bad %d'
          % (count, n, mode, alntype, band, runs[0][0][:34], runs[0][3], runs[1][0][:34], runs[1][3], 'same' if same else 'DIFFER', nbad), flush=True)
print('TOTAL BAD', bad)
sys.exit(1 if bad else 0)
