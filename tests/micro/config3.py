"""BASELINE config 3: full unbanded Smith-Waterman on ONE pair (default 100 kb x ~100 kb, 10 % divergence,
match 1 / mismatch -3 / go -5 / ge -2) through the tiled kernel.  No oracle exists at this size (the reference
would need ~0.5 TB): the check is size-independent -- re-scoring the transcript with the reference's rule
(a gap-open charge for every maximal gap run) reproduces the reported score, the path is consistent with the
reported start / end cells, and every M/S matches the letters.  With --oracle the result is also compared with
the CPU oracle (feasible up to ~20 kb).

    python tests/micro/config3.py [length] [--oracle]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from biseqt_amd import synth, _pwlib as W          # noqa: E402
from biseqt_amd.batch import BatchAligner          # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 100000
rng = synth.rng_for(3)
o = synth.rand_seqs(rng, 1, n)[0]
m = synth.mutate(rng, o, 0.07, 0.015, 0.5)
t0 = time.time()
b = BatchAligner([(o, m)], alnmode=0, alntype=1, alphabet_len=4, match_score=1, mismatch_score=-3, go_score=-5,
                 ge_score=-2, flags=W.PW_FLAG_PROFILE)
t1 = time.time()
b.solve(); b.traceback(); b.sync()
t2 = time.time()
b.solve(); b.traceback(); b.sync()
fill, trace = b.fill_ms(), b.trace_ms()
res = b.results()
tx = b.transcripts(res)[0]
cells = b.cells
print('kernel: %s   X=%d Y=%d cells=%.4g' % (b.kernel_name, len(o), len(m), cells))
print('create %.2f s, first run %.3f s; fill %.2f ms = %.1f GCUPS, traceback %.2f ms' % (t1 - t0, t2 - t1, fill, cells / fill / 1e6, trace))
score, i, j = 0, int(res['origin_idx'][0]), int(res['mutant_idx'][0])
ops = np.frombuffer(tx.encode(), dtype=np.uint8)
prev = 0
for op in ops:
    if op in (77, 83):                      # M, S
        assert (o[i] == m[j]) == (op == 77)
        score += 1 if op == 77 else -3
        i += 1; j += 1
    else:
        score += -2 + (-5 if op != prev else 0)
        if op == 68:
            i += 1
        else:
            j += 1
    prev = op
print('score %d, re-scored %d, transcript %d ops, start (%d,%d) end (%d,%d) reported end (%d,%d)'
      % (res['score'][0], score, len(tx), res['origin_idx'][0], res['mutant_idx'][0], i, j, res['opt_i'][0], res['opt_j'][0]))
assert score == res['score'][0] and (i, j) == (res['opt_i'][0], res['opt_j'][0])
if '--oracle' in sys.argv:
    from oracle import oracle as O
    t = time.time()
    r = O.solve(o, m, L=4, alntype=O.LOCAL, match=1, mismatch=-3, go=-5, ge=-2)
    print('oracle %.1f s' % (time.time() - t))
    assert r['score'] == res['score'][0] and r['opt'] == (res['opt_i'][0], res['opt_j'][0]) and r['transcript'] == tx
    print('matches the oracle: score, end cell, transcript')
print('OK')
b.close()
