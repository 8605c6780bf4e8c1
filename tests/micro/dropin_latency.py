"""Latency of the four drop-in calls (dptable_init / solve / traceback / free) for ONE pair, as the reference's
Aligner drives them, next to the compiled reference on this host's CPU -- and GCUPS of the batch path for the
non-headline kernel variants (32-bit, f64, generic, linear gaps) on the config-2 batch.

    python tests/micro/dropin_latency.py
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from biseqt_amd import synth, _pwlib as W          # noqa: E402
from biseqt_amd.batch import BatchAligner          # noqa: E402
from oracle import ref_driver as R                 # noqa: E402  (checker / baseline only)

SC = dict(match=1., mismatch=-3., go=-5., ge=-2.)


def once(lib, P, reps):
    R.run(lib, P)
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = R.run(lib, P)
        t.append(time.perf_counter() - t0)
    return out, float(np.median(t)) * 1e3


def main():
    ours = R.load(W.PWLIB_SO)
    ref = R.load() if os.path.exists(os.path.join(ROOT, 'oracle', '_ref', 'pwlib_ref.so')) else None
    rng = synth.rng_for(1)
    a, b = synth.rand_seqs(rng, 2, 1000)
    o2, m2 = synth.pair_batch(2, 1, 2000)
    o5 = synth.rand_seqs(rng, 1, 5000)[0]
    m5 = synth.mutate(rng, o5, 0.05, 0.03, 0.4)
    cases = [
        ('cfg1  1 kb x 1 kb STD GLOBAL 1/0/0/0', R.Problem(a.tolist(), b.tolist(), mode=0, alntype=0, L=4, match=1., mismatch=0., go=0., ge=0.)),
        ('cfg1  1 kb x 1 kb STD GLOBAL 1/-3/-5/-2', R.Problem(a.tolist(), b.tolist(), mode=0, alntype=0, L=4, **SC)),
        ('cfg2  one 2 kb pair, radius 200 B_LOCAL', R.Problem(o2[0].tolist(), m2[0].tolist(), mode=1, alntype=1, diag_range=(-200, 200), L=4, **SC)),
        ('5 kb pair, radius 300 B_OVERLAP', R.Problem(o5.tolist(), m5.tolist(), mode=1, alntype=2, diag_range=(-300, 300), L=4, **SC)),
    ]
    devnull = os.open(os.devnull, os.O_WRONLY)
    saved = os.dup(1)
    rows = []
    for name, P in cases:
        os.dup2(devnull, 1)
        g, tg = once(ours, P, 20)
        if ref is not None:
            c, tc = once(ref, P, 3)
        os.dup2(saved, 1)
        same = ref is None or (g['opt'] == c['opt'] and g.get('transcript') == c.get('transcript') and g.get('score') == c.get('score'))
        rows.append((name, tg, tc if ref is not None else float('nan'), same))
    for name, tg, tc, same in rows:
        print('%-44s drop-in %8.3f ms   reference CPU %9.2f ms   x%-7.0f identical=%s' % (name, tg, tc, tc / tg, same))

    # batch variants on the config-2 batch
    origins, mutants = synth.pair_batch(2, 10000, 2000)
    pairs = list(zip(origins, mutants))
    base = dict(alnmode=1, alntype=1, alphabet_len=4, diag_range=(-200, 200), match_score=1, mismatch_score=-3,
                go_score=-5, ge_score=-2)
    variants = [('default (packed 16-bit)', {}), ('32-bit', dict(flags=W.PW_FLAG_NO_PACKED16)),
                ('f64', dict(flags=W.PW_FLAG_FORCE_F64)), ('generic (matrix in LDS)', dict(flags=W.PW_FLAG_FORCE_GENERIC)),
                ('linear gaps go=0', dict(go_score=0)), ('B_OVERLAP', dict(alntype=2)), ('B_GLOBAL (-200,200)', dict(alntype=0))]
    for name, kw in variants:
        k = dict(base); k.update(kw)
        k['flags'] = k.get('flags', 0) | W.PW_FLAG_PROFILE
        with BatchAligner(pairs, **k) as bt:
            bt.solve(); bt.traceback(); bt.sync()
            t0 = time.perf_counter()
            for _ in range(5):
                bt.solve(); bt.traceback()
            bt.sync()
            dt = (time.perf_counter() - t0) / 5
            print('cfg2 batch, %-26s %-22s fill %6.2f ms  trace %5.2f ms  end-to-end %7.1f GCUPS'
                  % (name, bt.kernel_name, bt.fill_ms(), bt.trace_ms(), bt.cells / dt / 1e9))


if __name__ == '__main__':
    main()
