"""Per-strip timeline of the strip pipeline (PWLIB_STRIP_TRACE): where a hop's time goes.

    python tests/micro/strip_trace.py X Y
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, 'gpurun_out', 'strip_trace.bin')
os.environ['PWLIB_STRIP_TRACE'] = out
from biseqt_amd import synth, _pwlib as W          # noqa: E402
from biseqt_amd.batch import BatchAligner          # noqa: E402

X, Y = int(sys.argv[1]), int(sys.argv[2])
rng = synth.rng_for(33)
o = synth.rand_seqs(rng, 1, X)[0]
m = synth.rand_seqs(rng, 1, Y)[0]
with BatchAligner([(o, m)], alnmode=0, alntype=1, alphabet_len=4, match_score=1, mismatch_score=-3, go_score=-5, ge_score=-2,
                  flags=W.PW_FLAG_PROFILE | W.PW_FLAG_FORCE_STRIP) as b:
    b.solve(); b.sync()
    b.solve(); b.sync()
    print('fill %.3f ms' % b.fill_ms())
t = np.fromfile(out, dtype=np.uint64).reshape(-1, 16)
t0 = t[:, 0].min()
us = (t[:, :7].astype(np.int64) - int(t0)) / 100.0
xcc = (t[:, 7] & 0xff).astype(int)
hw = (t[:, 7] >> 32).astype(np.int64)
wg = ((t[:, 7] >> 8) & 0xffffff).astype(int)
# HW_REG_HW_ID (gfx9 layout): wave slot [3:0], SIMD [5:4], CU [11:8], SH [12], SE [15:13]
place = np.stack([xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, (hw >> 4) & 3], axis=1)
workers = {}
for w in range(len(t)):
    workers.setdefault(wg[w], tuple(place[w]))
simds = {}
for g, pl in workers.items():
    simds.setdefault(pl, []).append(g)
cus = {}
for pl in simds:
    cus.setdefault(pl[:4], []).append(pl[4])
print('workers seen %d on %d distinct SIMDs in %d CUs; SIMDs with more than one worker: %d (max %d); workers per CU: %s'
      % (len(workers), len(simds), len(cus), sum(1 for v in simds.values() if len(v) > 1), max(len(v) for v in simds.values()),
         dict(zip(*np.unique([sum(len(simds[c + (s_,)]) for s_ in set(v)) for c, v in cus.items()], return_counts=True)))))
names = ['dequeue', 'setup', 'granules seen', 'step 64', 'step 96', 'end', 'first block past the steady ones']
print('strips %d; columns: %s (us since the first dequeue)' % (len(t), ', '.join(names)))
for w in list(range(0, min(len(t), 12))) + list(range(124, min(len(t), 134))) + list(range(len(t) - 3, len(t))):
    if 0 <= w < len(t):
        print('strip %5d xcc %d  ' % (w, xcc[w]) + '  '.join('%9.2f' % v for v in us[w]))
d = np.diff(us[:, 2])
print('hop (granules seen, w -> w+1): median %.2f us, mean %.2f, p90 %.2f; same-XCD hops %.2f, cross-XCD %.2f'
      % (np.median(d), d.mean(), np.percentile(d, 90), np.median(d[xcc[1:] == xcc[:-1]]) if (xcc[1:] == xcc[:-1]).any() else -1,
         np.median(d[xcc[1:] != xcc[:-1]]) if (xcc[1:] != xcc[:-1]).any() else -1))
nst = (Y + 64 + 31) // 32 * 32
first_end = ((Y - 63) // 32 + 1) * 32 if Y >= 63 else 0          # first block past the steady ones (k0 + 63 > Y)
print('steady part (step 96 -> first end block at step %d): %.1f ns per step; end part (%d steps): median %.2f us = %.1f ns per step'
      % (first_end, 1000 * np.median(us[:, 6] - us[:, 4]) / max(1, first_end - 96), nst - first_end,
         np.median(us[:, 5] - us[:, 6]), 1000 * np.median(us[:, 5] - us[:, 6]) / max(1, nst - first_end)))
# the shader clock each strip ran at: counts between "granules seen" [2] and "end" [5] over the 100 MHz time between them
cyc = (t[:, 13].astype(np.int64) - t[:, 10].astype(np.int64)).astype(np.float64)
dur = (t[:, 5].astype(np.int64) - t[:, 2].astype(np.int64)).astype(np.float64) / 100.0        # us
ok = dur > 0
mhz = cyc[ok] / dur[ok]
ws = np.nonzero(ok)[0]
print('shader clock per strip (counts / time, MHz): strip 0 %.0f, median %.0f, min %.0f, max %.0f; strips %s: %s'
      % (mhz[0], np.median(mhz), mhz.min(), mhz.max(), [int(ws[i]) for i in (len(ws) // 4, len(ws) // 2, -1)],
         ['%.0f' % mhz[i] for i in (len(ws) // 4, len(ws) // 2, -1)]))
print('strip duration (granules seen -> end, us): strip 0 %.0f, strip %d %.0f, last %.0f; in shader cycles per step: %.1f, %.1f, %.1f'
      % (dur[0], len(dur) // 2, dur[len(dur) // 2], dur[-1], cyc[0] / (Y + 64), cyc[len(dur) // 2] / (Y + 64), cyc[-1] / (Y + 64)))
polls = (t[:, 15] & 0xffffffff).astype(np.int64); spins = (t[:, 15] >> 32).astype(np.int64)
print('hand-overs that had to poll, per strip (of %d): median %d, mean %.1f, max %d; polls per such hand-over: %.1f; strips %s: %s'
      % ((Y + 64) // 16, np.median(polls), polls.mean(), polls.max(), spins.sum() / max(1, polls.sum()),
         [0, 1, 100, len(t) // 2, len(t) - 1], [int(polls[i]) for i in (0, 1, 100, len(t) // 2, len(t) - 1)]))
print('granules seen -> step 64: median %.2f us; step 64 -> 96: %.2f us; strip total: %.2f us'
      % (np.median(us[:, 3] - us[:, 2]), np.median(us[:, 4] - us[:, 3]), np.median(us[:, 5] - us[:, 2])))
