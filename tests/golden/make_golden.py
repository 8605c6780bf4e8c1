"""Generate the golden vectors under tests/golden/ from the COMPILED REFERENCE.

    python tests/golden/make_golden.py          (in the build container, where /root/reference exists)

Every expected value below is an output of oracle/_ref/pwlib_ref.so, i.e. of the reference's own C
sources compiled unmodified by oracle/Makefile, driven through its four ABI functions by
oracle/ref_driver.py.  Fixtures are data only: inputs + the reference's outputs.

Files written:
  known_answers.json   the known-answer cases of the reference's tests/test_pw.py:33-103 and the
                       pw.py:11-21 docstring example, plus the non-Gotoh witness of SURVEY.md section 7
  random_matrix.json.gz  random problems: {7 STD types, 3 banded types} x go {<0,0,>0} x ge x |alphabet|
                       x lengths 0..64 x random bands (clamped / infeasible) x sub-frames
  float_logodds.json   log-odds float scores (the reference's own MutationProcess.log_odds_scores, stochastics.py:234-310) at the five noise
                       levels of tests/test_pw.py:106; scores stored as exact hex floats
  config_sized.json    config-sized spot checks: 1 kb global, 2 kb r=200 B_LOCAL pairs, 5 kb B_OVERLAP

Inputs on which the reference's traceback would exit(1) (pw.c:132-134) are recognised beforehand
with oracle/pw_oracle.c (`would_panick`) and recorded with `panick: true` and no transcript.
"""
import gzip
import hashlib
import json
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import oracle as O          # noqa: E402
from oracle import ref_driver as R      # noqa: E402
from biseqt_amd import synth            # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
REF = R.load()


def enc(seq):
    return ''.join(chr(ord('0') + int(c)) for c in seq)


def reference_record(origin, mutant, kw, keep_transcript=True, name=None):
    o = O.solve(origin, mutant, **kw)
    panick = bool(o.get('would_panick'))
    r = R.run(REF, R.Problem(origin, mutant, **kw), do_traceback=not panick)
    exp = dict(init_rc=r['init_rc'])
    if 'band' in r:
        exp['band'] = list(r['band'])
    if r['init_rc'] == 0:
        exp['opt'] = list(r['opt'])
        exp['num_rows'] = r['num_rows']
        if r['score'] is not None:
            exp['score'] = r['score']
            exp['score_hex'] = float(r['score']).hex()
            exp['panick'] = panick
            if not panick:
                exp['tb_null'] = r['tb_null']
                if not r['tb_null']:
                    tx = r['transcript']
                    exp['origin_idx'] = r['origin_idx']
                    exp['mutant_idx'] = r['mutant_idx']
                    exp['tx_len'] = len(tx)
                    exp['tx_sha256'] = hashlib.sha256(tx.encode()).hexdigest()
                    if keep_transcript:
                        exp['transcript'] = tx
    rec = dict(origin=enc(origin), mutant=enc(mutant), kw=kw, expect=exp)
    if name:
        rec['name'] = name
    return rec


def known_answers():
    recs = []
    A = 'ACGT'
    e = lambda s: [A.index(c) for c in s]   # noqa: E731
    recs.append(reference_record(e('AAACGCGT'), e('AACGCCTT'), dict(L=4, mode=0, alntype=0),
                                 name='pw.py:11-21 docstring example'))
    for L, tag in ((4, 'one-letter alphabet'), (2, 'two-letter alphabet')):
        S = [0] * 10
        junk = [1] * 10
        recs.append(reference_record(S, S, dict(L=L, mode=0, alntype=0),
                                     name='test_pw.py:33-42 global self (%s)' % tag))
        recs.append(reference_record(S, S[:5], dict(L=L, mode=0, alntype=0),
                                     name='test_pw.py:44-50 global with gaps (%s)' % tag))
        recs.append(reference_record(S + junk, junk + S, dict(L=L, mode=0, alntype=R.LOCAL),
                                     name='test_pw.py:52-58 local (%s)' % tag))
        recs.append(reference_record(S, junk, dict(L=L, mode=0, alntype=R.LOCAL),
                                     name='test_pw.py:60-62 local not found (%s)' % tag))
        recs.append(reference_record(S + junk, junk + S, dict(L=L, mode=0, alntype=R.OVERLAP),
                                     name='test_pw.py:64-67 overlap (%s)' % tag))
        recs.append(reference_record(S, S, dict(L=L, mode=1, alntype=R.B_GLOBAL, diag_range=(0, 0)),
                                     name='test_pw.py:80-83 banded global (%s)' % tag))
        recs.append(reference_record(S + junk, junk + S,
                                     dict(L=L, mode=1, alntype=R.B_OVERLAP, diag_range=(-20, 20),
                                          ge=-1.),
                                     name='test_pw.py:85-92 banded overlap (%s)' % tag))
    # test_pw.py:95-103 memory test, scaled to 20000 (the 1e6 original is run live in the gpu tests
    # as a property: transcript == 'S' * L)
    n = 20000
    recs.append(reference_record([0] * n, [1] * n, dict(L=4, mode=1, alntype=0, diag_range=(0, 0)),
                                 keep_transcript=False,
                                 name='test_pw.py:95-103 banded memory (scaled to 2e4)'))
    # non-Gotoh witness, SURVEY.md section 7 "The reference is not Gotoh"
    recs.append(reference_record([2, 0, 2, 0, 3, 3, 0], [0, 0, 0],
                                 dict(L=4, mode=0, alntype=0, match=2., mismatch=-3., go=-4., ge=-1.),
                                 name='non-Gotoh witness (reference -11, textbook affine -7)'))
    return recs


def random_matrix(n=3000, seed=12345):
    rng = np.random.default_rng(seed)
    recs = []
    std_types = list(range(7))
    b_types = list(range(3))
    for t in range(n):
        L = int(rng.choice([2, 4]))
        if t % 5 == 0:
            nn, mm = int(rng.integers(0, 7)), int(rng.integers(0, 7))      # tiny / empty
        else:
            nn, mm = int(rng.integers(1, 65)), int(rng.integers(1, 65))
        origin = rng.integers(0, L, nn).tolist()
        if nn and rng.random() < 0.6:
            mutant = synth.mutate(rng, np.array(origin, np.uint8), 0.1, 0.08, 0.3, L).tolist()
            if rng.random() < 0.3:
                mutant = rng.integers(0, L, int(rng.integers(0, 9))).tolist() + mutant
            mm = len(mutant)
        else:
            mutant = rng.integers(0, L, mm).tolist()
        kw = dict(L=L)
        if rng.random() < 0.2:
            kw['subst'] = rng.integers(-4, 5, (L, L)).astype(float).tolist()
        else:
            kw['match'] = float(rng.choice([1, 2, 5]))
            kw['mismatch'] = float(rng.choice([0, -1, -3]))
        kw['go'] = float([0, -1, -5, 3][t % 4] if t % 7 else rng.choice([0, -4, 2]))
        kw['ge'] = float(rng.choice([0, -1, -2]))
        if rng.random() < 0.25 and nn and mm:
            a, b = sorted(rng.integers(0, nn + 1, 2).tolist())
            kw['origin_range'] = [a, b]
            a, b = sorted(rng.integers(0, mm + 1, 2).tolist())
            kw['mutant_range'] = [a, b]
        if t % 2 == 0:
            kw['mode'] = 0
            kw['alntype'] = std_types[(t // 2) % 7]
        else:
            kw['mode'] = 1
            kw['alntype'] = b_types[(t // 2) % 3]
            X = (kw['origin_range'][1] - kw['origin_range'][0]) if 'origin_range' in kw else nn
            Y = (kw['mutant_range'][1] - kw['mutant_range'][0]) if 'mutant_range' in kw else mm
            r = rng.random()
            if r < 0.5:      # band around the end-point diagonal
                c = X - Y
                w = int(rng.integers(0, 12))
                lo, hi = min(c, 0) - w, max(c, 0) + w
            elif r < 0.8:    # arbitrary, may be clamped or infeasible for B_GLOBAL
                lo, hi = sorted(rng.integers(-Y - 4, X + 5, 2).tolist())
            else:            # one-sided
                lo = int(rng.integers(1, 6))
                hi = lo + int(rng.integers(0, 8))
                if rng.random() < 0.5:
                    lo, hi = -hi, -lo
            kw['diag_range'] = [int(lo), int(hi)]
        recs.append(reference_record(origin, mutant, kw))
    return recs


_REF_STOCHASTICS = None


def log_odds(err, go_prob=None):
    """Log-odds scores from the reference's OWN `MutationProcess.log_odds_scores`
    (`/root/reference/biseqt/stochastics.py:234-310`, imported here -- it runs under python 3 unchanged; nothing is written
    next to it) for subst_probs = ge_prob = err over ACGT, uniform null hypothesis; go_prob = err unless given."""
    global _REF_STOCHASTICS
    if _REF_STOCHASTICS is None:
        sys.dont_write_bytecode = True
        sys.path.insert(0, '/root/reference')
        import biseqt.sequence as RS
        import biseqt.stochastics as RT
        _REF_STOCHASTICS = (RS, RT)
    RS, RT = _REF_STOCHASTICS
    M = RT.MutationProcess(RS.Alphabet('ACGT'), subst_probs=err, go_prob=err if go_prob is None else go_prob, ge_prob=err)
    S, (go, ge) = M.log_odds_scores()
    return [[float(v) for v in row] for row in S], float(go), float(ge)


def float_logodds(seed=777):
    rng = np.random.default_rng(seed)
    recs = []
    for err in (1e-2, 1e-1, 2e-1, 3e-1, 4e-1):
        S, go, ge = log_odds(err)
        # an affine variant too (go_prob < ge_prob), which tests/test_pw.py never exercises
        go2 = log_odds(err, go_prob=err / 2)[1]
        for rep in range(6):
            origin = rng.integers(0, 4, 100).tolist()
            mutant = synth.mutate(rng, np.array(origin, np.uint8), err, err, err, 4).tolist()
            for (mode, typ, extra) in ((0, R.GLOBAL, {}), (0, R.LOCAL, {'pad': True}),
                                       (1, R.B_OVERLAP, {'diag_range': [-30, 30]})):
                mut = mutant
                if extra.get('pad'):
                    mut = [0] * 100 + mutant + [2] * 100       # test_pw.py:166
                kw = dict(L=4, subst=S, go=(go if rep % 2 == 0 else go2), ge=ge, mode=mode,
                          alntype=typ)
                if 'diag_range' in extra:
                    kw['diag_range'] = extra['diag_range']
                rec = reference_record(origin, mut, kw)
                rec['err'] = err
                rec['kw_hex'] = dict(subst=[[float(v).hex() for v in row] for row in S],
                                     go=float(kw['go']).hex(), ge=float(ge).hex())
                recs.append(rec)
    return recs


def config_sized():
    recs = []
    rng = synth.rng_for(1)
    a, b = synth.rand_seqs(rng, 2, 1000)
    for sc in (dict(), dict(match=1., mismatch=-3., go=-5., ge=-2.)):
        recs.append(reference_record(a.tolist(), b.tolist(), dict(L=4, mode=0, alntype=0, **sc),
                                     keep_transcript=False, name='cfg1 1kb x 1kb STD GLOBAL'))
    origins, mutants = synth.pair_batch(2, 6, 2000)
    for k in range(6):
        for sc in (dict(match=1., mismatch=-3., go=-5., ge=-2.), dict(match=1., mismatch=-3., go=0., ge=-2.)):
            recs.append(reference_record(origins[k].tolist(), mutants[k].tolist(),
                                         dict(L=4, mode=1, alntype=R.B_LOCAL, diag_range=[-200, 200], **sc),
                                         keep_transcript=False, name='cfg2 unit 2kb r=200 B_LOCAL'))
    rng = synth.rng_for(4)
    g = synth.rand_seqs(rng, 1, 8000)[0]
    r1 = g[:5000]
    r2 = synth.mutate(rng, g[3000:8000], 0.05, 0.05, 0.3)
    recs.append(reference_record(r1.tolist(), r2.tolist(),
                                 dict(L=4, mode=1, alntype=R.B_OVERLAP, diag_range=[2800, 3200],
                                      match=1., mismatch=-3., go=-5., ge=-2.),
                                 keep_transcript=False, name='cfg4 unit 5kb B_OVERLAP r=200'))
    # a moderately sized STD LOCAL (config-3 shape in miniature)
    o = synth.rand_seqs(rng, 1, 700)[0]
    m = synth.mutate(rng, o, 0.1, 0.05, 0.3)
    recs.append(reference_record(o.tolist(), m.tolist(),
                                 dict(L=4, mode=0, alntype=R.LOCAL, match=1., mismatch=-3., go=-5., ge=-2.),
                                 keep_transcript=False, name='cfg3 miniature 700 x ~700 STD LOCAL'))
    return recs


def dump(name, recs, gz=False):
    path = os.path.join(HERE, name)
    txt = json.dumps(dict(generator='tests/golden/make_golden.py',
                          source='oracle/_ref/pwlib_ref.so (reference C sources compiled unmodified)',
                          records=recs), separators=(',', ':'))
    if gz:
        with gzip.GzipFile(path, 'wb', mtime=0) as f:
            f.write(txt.encode())
    else:
        with open(path, 'w') as f:
            f.write(txt)
    print(name, len(recs), 'records', os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    dump('known_answers.json', known_answers())
    dump('random_matrix.json.gz', random_matrix(), gz=True)
    dump('float_logodds.json', float_logodds())
    dump('config_sized.json', config_sized())
