"""Is the planner's pick still the fastest?  For a list of shapes the batch is run the way the planner chooses and the ways
it rejected (forcing flags / knobs), fill + traceback timed by the library's HIP events; a pick more than `tolerance` slower
than the best alternative is reported.  The timing model behind the picks is biseqt_amd/csrc/pw_model.h: when a kernel gets
faster (or slower) its constants go stale -- this script (all shapes) and tests/test_gpu_planner.py (a bounded subset) say so.

    python tests/micro/planner_check.py [--quick]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from biseqt_amd import synth, _pwlib as W          # noqa: E402
from biseqt_amd.batch import BatchAligner          # noqa: E402

KNOBS = ('PWLIB_NO_SMALL_STRIP', 'PWLIB_NO_PACKED_MW', 'PWLIB_NO_STRIP', 'PWLIB_NO_SMALL_TILED', 'PWLIB_LATENCY_MODE',
         'PWLIB_NO_SCALED16', 'PWLIB_PACKED_BK')
CFG = dict(match_score=1, mismatch_score=-3, go_score=-5, ge_score=-2)
F64 = dict(match_score=0.3, mismatch_score=-1.1, go_score=-2.2, ge_score=-0.7)

# (title, n pairs, length, standard mode?, alntype, band radius or None, scores, alternatives as (name, flags, env))
ALT_STD = [('32-bit workgroups', 0, {'PWLIB_NO_SMALL_STRIP': '1', 'PWLIB_NO_PACKED_MW': '1'}),
           ('packed workgroups', 0, {'PWLIB_NO_SMALL_STRIP': '1'}),
           ('strips', W.PW_FLAG_FORCE_STRIP, {})]
ALT_F64 = [('workgroups', 0, {'PWLIB_NO_SMALL_TILED': '1'}), ('tiles', W.PW_FLAG_FORCE_TILED, {})]
ALT_BAND = [('throughput layout', 0, {'PWLIB_LATENCY_MODE': '0'}), ('latency layout', 0, {'PWLIB_LATENCY_MODE': '1'}),
            ('32-bit', W.PW_FLAG_NO_PACKED16, {})]
SHAPES = [
    ('one 500 x 500 LOCAL', 1, 500, True, 1, None, CFG, ALT_STD),
    ('four 500 x 500 LOCAL', 4, 500, True, 1, None, CFG, ALT_STD),
    ('one 2 kb x 2 kb LOCAL', 1, 2000, True, 1, None, CFG, ALT_STD),
    ('four 2 kb x 2 kb LOCAL', 4, 2000, True, 1, None, CFG, ALT_STD),
    ('sixteen 1.2 kb x 1.2 kb LOCAL', 16, 1200, True, 1, None, CFG, ALT_STD),
    ('one 4 kb x 4 kb LOCAL', 1, 4000, True, 1, None, CFG, ALT_STD),
    ('four 4 kb x 4 kb LOCAL', 4, 4000, True, 1, None, CFG, ALT_STD),
    ('one 8 kb x 8 kb GLOBAL', 1, 8000, True, 0, None, CFG, ALT_STD),
    ('one 6 kb x 6 kb LOCAL, f64 scores', 1, 6000, True, 1, None, F64, ALT_F64),
    ('eight 3 kb x 3 kb LOCAL, f64 scores', 8, 3000, True, 1, None, F64, ALT_F64),
    ('16 pairs of 2 kb, band radius 20', 16, 2000, False, 1, 20, CFG, ALT_BAND),
    ('200 pairs of 2 kb, band radius 50', 200, 2000, False, 1, 50, CFG, ALT_BAND),
    ('2000 pairs of 1 kb, band radius 10', 2000, 1000, False, 1, 10, CFG, ALT_BAND),
    ('10 000 pairs of 2 kb, band radius 200 (config 2)', 10000, 2000, False, 1, 200, CFG, ALT_BAND),
]
QUICK = (0, 2, 3, 5, 6, 8, 10, 12)


def _time(pairs, kw, flags, env, reps=3):
    for k in KNOBS:
        os.environ.pop(k, None)
    os.environ.update(env)
    try:
        with BatchAligner(pairs, flags=flags | W.PW_FLAG_PROFILE, **kw) as b:
            ts = []
            for _ in range(reps):
                b.solve(); b.traceback(); b.sync()
                ts.append(b.fill_ms() + b.trace_ms())
            return min(ts), b.kernel_name
    finally:
        for k in env:
            os.environ.pop(k, None)


def run(indices=None, tolerance=1.25, out=sys.stdout):
    """Returns the list of (title, pick ms, pick kernel, best alternative ms, its name) for picks slower than tolerance x best."""
    rng = synth.rng_for(909)
    late = []
    for idx, (title, count, n, std, alntype, radius, scores, alts) in enumerate(SHAPES):
        if indices is not None and idx not in indices:
            continue
        pairs = []
        for _ in range(count):
            o = synth.rand_seqs(rng, 1, n)[0]
            pairs.append((o, synth.mutate(rng, o, 0.07, 0.02, 0.4)))
        kw = dict(alnmode=0 if std else 1, alntype=alntype, alphabet_len=4, **scores)
        if not std:
            kw['diag_range'] = (-radius, radius)
        pick_ms, pick_kernel = _time(pairs, kw, 0, {})
        rows = []
        for name, flags, env in alts:
            try:
                ms, kernel = _time(pairs, kw, flags, env)
            except RuntimeError as e:               # an alternative the library refuses for this shape
                rows.append((name, None, str(e)[:40]))
                continue
            rows.append((name, ms, kernel))
        best = min((r for r in rows if r[1] is not None), key=lambda r: r[1])
        verdict = 'ok' if pick_ms <= tolerance * best[1] else 'SLOW'
        out.write('%-50s planner %8.3f ms (%s)   %s   [%s]\n' % (
            title, pick_ms, pick_kernel[:30], '   '.join('%s %s' % (r[0], '%.3f ms' % r[1] if r[1] is not None else 'n/a') for r in rows), verdict))
        out.flush()
        if verdict != 'ok':
            late.append((title, pick_ms, pick_kernel, best[1], best[0]))
    return late


if __name__ == '__main__':
    late = run(QUICK if '--quick' in sys.argv else None)
    print('%d pick(s) more than 25 %% slower than an alternative' % len(late))
    sys.exit(1 if late else 0)
