"""Replays a pair the fuzz saved (gpurun_out/fuzz_bad_N.npz or tests/golden/regress/*.npz) under several kernel selections
and compares each with the oracle.    python tests/micro/replay_bad.py <file.npz>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from biseqt_amd import _pwlib as W          # noqa: E402
from biseqt_amd.batch import BatchAligner          # noqa: E402
from oracle import oracle as O                     # noqa: E402

d = np.load(sys.argv[1])
o, m = d['o'], d['m']
mode, alntype, flags0, L = (int(v) for v in d['meta'])
sc = [float(v) for v in d['sc']]
band = tuple(int(v) for v in d['band'])
okw = dict(L=L, mode=mode, alntype=alntype, match=sc[0], mismatch=sc[1], go=sc[2], ge=sc[3])
if mode == 1:
    okw['diag_range'] = band
r = O.solve(o, m, **okw)
print('oracle: opt', r['opt'], 'score', r['score'], 'tx', len(r['transcript'] or ''), 'start', r['origin_idx'], r['mutant_idx'])
for title, flags, env in (('default', flags0, {}), ('32-bit', W.PW_FLAG_NO_PACKED16, {}), ('no packed mw', 0, {'PWLIB_NO_PACKED_MW': '1'}),
                          ('f64', W.PW_FLAG_FORCE_F64, {}), ('tiled', W.PW_FLAG_FORCE_TILED, {}), ('throughput', 0, {'PWLIB_LATENCY_MODE': '0'}),
                          ('x3 pairs', 0, {})):
    os.environ.update(env)
    try:
        kw = dict(alnmode=mode, alntype=alntype, alphabet_len=L, match_score=sc[0], mismatch_score=sc[1], go_score=sc[2], ge_score=sc[3],
                  flags=flags, check_band=False)
        pairs = [(o, m)] * (3 if title == 'x3 pairs' else 1)
        if mode == 1:
            kw['diag_range'] = [band] * len(pairs)
        with BatchAligner(pairs, **kw) as b:
            res = b.run()
            txs = b.transcripts(res)
            ok = (int(res['opt_i'][0]), int(res['opt_j'][0])) == tuple(r['opt']) and res['score'][0] == r['score'] and txs[0] == r['transcript']
            print('%-14s %-44s opt (%d, %d) score %s  %s' % (title, b.kernel_name, res['opt_i'][0], res['opt_j'][0], res['score'][0], 'ok' if ok else 'MISMATCH'), flush=True)
    finally:
        for k in env:
            os.environ.pop(k, None)
