"""Parity of the HIP path (through the C ABI of pwlib.so) with the reference: the golden vectors
generated from the compiled reference, the oracle on seeded random inputs, and size-independent
properties at BASELINE config sizes.  Bit-exact for integer scores; for the floating-point log-odds
scores the comparison is also exact (same IEEE double additions in the same association), which is
stricter than the 3-decimals the reference's own tests ask for (tests/test_pw.py:126-135)."""
import collections
import hashlib
import json
import os

import numpy as np
import pytest

from tests.helpers import check_against_expect, dec, kw_of, load_golden

pytestmark = pytest.mark.gpu


def _loaded_native():
    """The round-end harness records which .so files were loaded: make sure ours is."""
    return any('pwlib/pwlib.so' in l for l in open('/proc/self/maps'))


def _frames(rec, kw):
    o, m = dec(rec['origin']), dec(rec['mutant'])
    orr = kw.get('origin_range', (0, len(o)))
    mrr = kw.get('mutant_range', (0, len(m)))
    return o[orr[0]:orr[1]], m[mrr[0]:mrr[1]], orr, mrr


def _group_key(kw):
    return json.dumps([kw['mode'], kw['alntype'], kw['L'], kw.get('subst'), kw.get('match'),
                       kw.get('mismatch'), kw['go'], kw['ge']])


def _run_group(recs_kws, flags=0):
    """Run golden records that share scoring/type as ONE batch; yields (index, got-dict)."""
    from biseqt_amd import _pwlib as W
    from biseqt_amd.batch import BatchAligner
    kw0 = recs_kws[0][2]
    pairs, drs, meta = [], [], []
    for idx, rec, kw in recs_kws:
        o, m, orr, mrr = _frames(rec, kw)
        pairs.append((np.array(o, np.uint8), np.array(m, np.uint8)))
        drs.append(tuple(kw.get('diag_range', (0, 0))))
        meta.append((idx, orr, mrr))
    bkw = dict(alnmode=kw0['mode'], alntype=kw0['alntype'], alphabet_len=kw0['L'], go_score=kw0['go'],
               ge_score=kw0['ge'], flags=flags)
    if kw0.get('subst') is not None:
        bkw['subst_scores'] = kw0['subst']
    else:
        bkw['match_score'], bkw['mismatch_score'] = kw0['match'], kw0['mismatch']
    # golden bands include out-of-table ones (clamped / rejected by the C side like dptable_init):
    # skip the Python-side assertion of pw.py:224-226
    if kw0['mode'] == 1:
        bkw['diag_range'] = drs
    b = BatchAligner(pairs, check_band=False, **bkw)
    with b:
        res = b.run()
        txs = b.transcripts(res)
        for k, (idx, orr, mrr) in enumerate(meta):
            got = dict(init_rc=b.init_rc(k), opt=None, score=None, transcript=None, origin_idx=None,
                       mutant_idx=None, tb_null=None, would_panick=None)
            if kw0['mode'] == 1:
                dmin, dmax, nrows = b.band(k)
                got['band'] = (dmin, dmax)
            if got['init_rc'] == 0:
                got['num_rows'] = b.band(k)[2]
                got['opt'] = (int(res['opt_i'][k]), int(res['opt_j'][k]))
                if got['opt'][0] != -1:
                    st = int(res['status'][k])
                    assert st & W.PW_ST_TRACED
                    got['score'] = float(res['score'][k])
                    got['would_panick'] = bool(st & W.PW_ST_PANICK)
                    got['tb_null'] = bool(st & W.PW_ST_EMPTY) and not (st & W.PW_ST_PANICK)
                    if not (st & (W.PW_ST_PANICK | W.PW_ST_EMPTY)):
                        got['transcript'] = txs[k]
                        got['origin_idx'] = int(res['origin_idx'][k]) + orr[0]
                        got['mutant_idx'] = int(res['mutant_idx'][k]) + mrr[0]
            yield idx, got


def _check_file(name, flags=0, every=1):
    recs = load_golden(name)
    groups = collections.OrderedDict()
    for idx in range(0, len(recs), every):
        kw = kw_of(recs[idx])
        groups.setdefault(_group_key(kw), []).append((idx, recs[idx], kw))
    n = 0
    for key, items in groups.items():
        for idx, got in _run_group(items, flags):
            check_against_expect(got, recs[idx]['expect'], where='%s[%d]' % (name, idx))
            n += 1
    assert _loaded_native()
    return n


def test_known_answers_batch():
    assert _check_file('known_answers.json') == 17


def test_random_matrix_batch_int32():
    assert _check_file('random_matrix.json.gz') == 3000


def test_random_matrix_batch_f64_and_generic():
    from biseqt_amd import _pwlib as W
    _check_file('random_matrix.json.gz', flags=W.PW_FLAG_FORCE_F64, every=2)
    _check_file('random_matrix.json.gz', flags=W.PW_FLAG_FORCE_GENERIC, every=3)
    _check_file('random_matrix.json.gz', flags=W.PW_FLAG_FORCE_GENERIC | W.PW_FLAG_FORCE_F64, every=5)


def test_float_logodds_batch():
    assert _check_file('float_logodds.json') == 90


def test_config_sized_batch():
    assert _check_file('config_sized.json') == 16


# ---- the reference's own caller-level tests (tests/test_pw.py) through Aligner -------------------
@pytest.mark.parametrize('letters', ['ACGT', ['00', '01']], ids=['one letter alphabet', 'two letter alphabet'])
def test_alignment_std_basic(letters):
    from biseqt_amd.pw import Aligner, Alignment, LOCAL, OVERLAP
    from biseqt_amd.sequence import Alphabet
    alphabet = Alphabet(letters)
    S = alphabet.parse(alphabet[0] * 10)
    with Aligner(S, S) as aligner:
        aligner.solve()
        assert aligner.traceback().transcript == 'M' * len(S)
        scores = aligner.table_scores()
        assert len(scores) == len(S) and all(len(row) == len(S) for row in scores)
        assert max(max(row) for row in scores) == scores[-1][-1]
    with Aligner(S, S[:len(S) // 2]) as aligner:
        aligner.solve()
        alignment = aligner.traceback()
        assert alignment.transcript.count('D') == len(S) // 2
        assert '-' * (len(S) // 2) in str(alignment)
    junk = alphabet.parse(alphabet[1] * len(S))
    origin, mutant = S + junk, junk + S
    alignment = Alignment(origin, mutant, 'M' * len(S), mutant_start=len(S))
    with Aligner(origin, mutant, alntype=LOCAL) as aligner:
        aligner.solve()
        assert alignment == aligner.traceback()
    with Aligner(S, junk, alntype=LOCAL) as aligner:
        assert aligner.solve() is None and aligner.traceback() is None
    with Aligner(origin, mutant, alntype=OVERLAP) as aligner:
        aligner.solve()
        assert aligner.traceback().transcript == 'M' * len(S)


@pytest.mark.parametrize('letters', ['ACGT', ['00', '01']], ids=['one letter alphabet', 'two letter alphabet'])
def test_alignment_banded_basic(letters):
    from biseqt_amd.pw import Aligner, Alignment, BANDED_MODE, B_OVERLAP
    from biseqt_amd.sequence import Alphabet
    alphabet = Alphabet(letters)
    S = alphabet.parse(alphabet[0] * 10)
    with Aligner(S, S, alnmode=BANDED_MODE, diag_range=(0, 0)) as aligner:
        aligner.solve()
        assert aligner.traceback() == Alignment(S, S, 'M' * len(S))
    junk = alphabet.parse(alphabet[1] * len(S))
    origin, mutant = S + junk, junk + S
    alignment = Alignment(origin, mutant, 'M' * len(S), mutant_start=len(S))
    with Aligner(origin, mutant, alnmode=BANDED_MODE, alntype=B_OVERLAP,
                 diag_range=(-2 * len(S), 2 * len(S)), ge_score=-1) as aligner:
        aligner.solve()
        assert alignment == aligner.traceback()


def test_alignment_banded_memory():
    """tests/test_pw.py:95-103 at full size: 1e6 x 1e6, band (0,0)."""
    from biseqt_amd.pw import Aligner, BANDED_MODE
    from biseqt_amd.sequence import Alphabet, Sequence
    A = Alphabet('ACGT')
    L = int(1e6)
    S, T = Sequence(A, (0,) * L), Sequence(A, (1,) * L)
    with Aligner(S, T, alnmode=BANDED_MODE, diag_range=(0, 0)) as aligner:
        aligner.solve()
        assert aligner.traceback().transcript == 'S' * L


@pytest.mark.parametrize('err', [1e-2, 1e-1, 2e-1, 3e-1, 4e-1])
@pytest.mark.parametrize('local', [False, True])
def test_alignment_log_odds_properties(err, local):
    """tests/test_pw.py:106-140 and :157-185 (float log-odds scores, 3-decimal tolerance as there)."""
    from biseqt_amd.pw import Aligner, Alignment, STD_MODE, GLOBAL, LOCAL
    from biseqt_amd.sequence import Alphabet
    from biseqt_amd.stochastics import MutationProcess, rand_seq
    rng = np.random.default_rng(int(err * 1000) + local)
    A = Alphabet('ACGT')
    M = MutationProcess(A, subst_probs=err, go_prob=err, ge_prob=err, rng=rng)
    subst_scores, (go_score, ge_score) = M.log_odds_scores()
    S = rand_seq(A, 100, rng=rng)
    T, tx = M.mutate(S)
    if local:
        T = A.parse('A' * 100) + T + A.parse('G' * 100)
        mutation_aln = Alignment(S, T, tx, mutant_start=100)
    else:
        mutation_aln = Alignment(S, T, tx)
    mutation_score = mutation_aln.calculate_score(subst_scores, go_score, ge_score)
    with Aligner(S, T, subst_scores=subst_scores, go_score=go_score, ge_score=ge_score,
                 alnmode=STD_MODE, alntype=LOCAL if local else GLOBAL) as aligner:
        reported_score = aligner.solve()
        assert round(reported_score, 3) >= round(mutation_score, 3)
        alignment = aligner.traceback()
        aln_score = alignment.calculate_score(subst_scores, go_score, ge_score)
        assert round(aln_score, 3) == round(reported_score, 3)
        assert round(aln_score, 3) == round(aligner.calculate_score(alignment), 3)
        ori_len = Alignment.projected_len(alignment.transcript, on='origin')
        mut_len = Alignment.projected_len(alignment.transcript, on='mutant')
        if local:
            assert ori_len <= len(S) and mut_len < len(T)
        else:
            assert ori_len == len(S) and mut_len == len(T)


def test_table_scores_match_oracle(oracle):
    from biseqt_amd.pw import Aligner, LOCAL
    from biseqt_amd.sequence import Alphabet, Sequence
    rng = np.random.default_rng(9)
    A = Alphabet('ACGT')
    o, m = rng.integers(0, 4, 70).tolist(), rng.integers(0, 4, 55).tolist()
    with Aligner(Sequence(A, o), Sequence(A, m), alntype=LOCAL, match_score=2, mismatch_score=-1,
                 go_score=-2, ge_score=-1) as aligner:
        score = aligner.solve()
        table = aligner.table_scores()
    ref = oracle.solve(o, m, L=4, alntype=oracle.LOCAL, match=2, mismatch=-1, go=-2, ge=-1, want_table=True)
    assert score == ref['score']
    H = ref['H'].reshape(71, 56)
    assert np.array_equal(np.array(table), H[:70, :55])


# ---- seeded random batches against the oracle, and properties at BASELINE sizes ------------------
def test_random_batch_against_oracle(oracle):
    from biseqt_amd import synth
    from biseqt_amd.batch import BatchAligner
    origins, mutants = synth.pair_batch(42, 96, 600)
    pairs = list(zip(origins, mutants))
    sc = dict(match_score=1, mismatch_score=-3, go_score=-5, ge_score=-2)
    for mode, typ, dr in ((1, 1, (-60, 60)), (1, 2, (-100, 40)), (0, 1, None), (0, 0, None)):
        kw = dict(alnmode=mode, alntype=typ, alphabet_len=4, **sc)
        if dr:
            kw['diag_range'] = dr
        with BatchAligner(pairs, **kw) as b:
            res = b.run()
            txs = b.transcripts(res)
        for k in range(0, len(pairs), 5):
            r = oracle.solve(origins[k], mutants[k], L=4, mode=mode, alntype=typ, diag_range=dr,
                             match=1, mismatch=-3, go=-5, ge=-2)
            assert (res['opt_i'][k], res['opt_j'][k]) == r['opt'], (mode, typ, k)
            assert res['score'][k] == r['score']
            assert txs[k] == r['transcript']
            assert (res['origin_idx'][k], res['mutant_idx'][k]) == (r['origin_idx'], r['mutant_idx'])


def _rescore(o, m, tx, i, j, match, mismatch, go, ge):
    s, prev = 0, ''
    for op in tx:
        if op in 'MS':
            s += match if o[i] == m[j] else mismatch
            assert (o[i] == m[j]) == (op == 'M')
            i, j = i + 1, j + 1
        else:
            s += ge + (go if op != prev else 0)
            if op == 'D':
                i += 1
            else:
                j += 1
        prev = op
    return s, i, j


def test_config2_properties_full_size(oracle):
    """BASELINE config 2 shape (2 kb x 2 kb, band radius 200, B_LOCAL, 1/-3/-5/-2) on 512 pairs:
    re-scoring every transcript reproduces its score, the path stays inside the band and ends at the
    reported end cell; a sample is compared with the oracle cell for cell."""
    from biseqt_amd import synth
    from biseqt_amd.batch import BatchAligner
    n = 512
    origins, mutants = synth.pair_batch(2, n, 2000)
    with BatchAligner(list(zip(origins, mutants)), alnmode=1, alntype=1, alphabet_len=4,
                      diag_range=(-200, 200), match_score=1, mismatch_score=-3, go_score=-5, ge_score=-2) as b:
        assert b.score_dtype == 'i32'
        # the kernel bench.py's headline line reports (scores held times 4; match / mismatch fed through the matrix form)
        assert b.kernel_name == 'k_fill16<8, false> x4 matrix'
        cells = b.cells
        res = b.run()
        txs = b.transcripts(res)
    assert cells == sum(synth.banded_cells(len(o), len(m), -200, 200) for o, m in zip(origins, mutants))
    for k in range(n):
        o, m = origins[k], mutants[k]
        s, i, j = _rescore(o, m, txs[k], int(res['origin_idx'][k]), int(res['mutant_idx'][k]), 1, -3, -5, -2)
        assert s == res['score'][k], k
        d = int(res['opt_i'][k]) - 200          # band not clamped at this size: dmin = -200
        a = int(res['opt_j'][k])
        assert (i, j) == (a + max(d, 0), a - min(d, 0)), k
    for k in range(0, n, 64):
        r = oracle.solve(origins[k], mutants[k], L=4, mode=1, alntype=1, diag_range=(-200, 200),
                         match=1, mismatch=-3, go=-5, ge=-2)
        assert (res['opt_i'][k], res['opt_j'][k]) == r['opt'] and res['score'][k] == r['score']
        assert txs[k] == r['transcript']


def test_config2_ten_thousand_pairs_every_kernel_agrees(oracle):
    """The batch bench.py times -- 10 000 pairs, 2 kb x ~2 kb, band radius 200, B_LOCAL, 1/-3/-5/-2 -- with the six
    config-sized golden pairs of that scoring embedded in it: the default run must use the kernel the bench reports,
    every transcript must re-score to its score and end in the reported cell, the golden pairs must equal the
    reference's answers, a sample must equal the oracle, and the 32-bit and f64 kernels must return the very same
    records and transcripts for all 10 000 pairs."""
    from biseqt_amd import _pwlib as W
    from biseqt_amd import synth, verify
    from biseqt_amd.batch import BatchAligner
    n = 10000
    origins, mutants = synth.pair_batch(2, n, 2000)
    gold = [(k, rec) for k, rec in enumerate(load_golden('config_sized.json'))
            if rec['kw'].get('alntype') == 1 and rec['kw'].get('mode') == 1 and rec['kw'].get('go') == -5.0]
    assert len(gold) == 6
    where = {}
    for q, (k, rec) in enumerate(gold):
        slot = 17 + 1613 * q
        origins[slot] = np.array(dec(rec['origin']), np.uint8)
        mutants[slot] = np.array(dec(rec['mutant']), np.uint8)
        where[slot] = (k, rec)
    pairs = list(zip(origins, mutants))
    kw = dict(alnmode=1, alntype=1, alphabet_len=4, diag_range=(-200, 200), match_score=1, mismatch_score=-3,
              go_score=-5, ge_score=-2)
    runs = {}
    for name, flags, env in (('packed16', 0, {}), ('packed16_plain', 0, {'PWLIB_SIMPLE_AS_MATRIX': '0'}),
                             ('packed16_unscaled', 0, {'PWLIB_NO_SCALED16': '1'}),
                             ('packed16_plain_unscaled', 0, {'PWLIB_NO_SCALED16': '1', 'PWLIB_SIMPLE_AS_MATRIX': '0'}),
                             ('int32', W.PW_FLAG_NO_PACKED16, {}), ('f64', W.PW_FLAG_FORCE_F64, {})):
        os.environ.update(env)
        try:
            with BatchAligner(pairs, flags=flags, **kw) as b:
                kname = b.kernel_name
                res = b.run()
                runs[name] = (kname, res.copy(), b.transcripts(res))
        finally:
            for key in env:
                os.environ.pop(key, None)
    assert runs['packed16'][0] == 'k_fill16<8, false> x4 matrix'
    assert runs['packed16_plain'][0] == 'k_fill16<8, false> x4'
    assert runs['packed16_unscaled'][0] == 'k_fill16<8, false> matrix'
    assert runs['packed16_plain_unscaled'][0] == 'k_fill16<8, false>'
    assert runs['int32'][0] == 'k_fill<int, 8, true, true, false>'
    assert runs['f64'][0] == 'k_fill<double, 8, true, true, false>'
    kname, res, txs = runs['packed16']
    assert (res['status'] == W.PW_ST_TRACED).all() and (res['opt_i'] >= 0).all()
    assert verify.check_batch(origins, mutants, res, txs, 1, -3, -5, -2, banded=True, dmins=[-200] * n) == []
    for slot, (k, rec) in where.items():
        e = rec['expect']
        assert (int(res['opt_i'][slot]), int(res['opt_j'][slot])) == tuple(e['opt']), k
        assert res['score'][slot] == e['score'] and len(txs[slot]) == e['tx_len'], k
        assert hashlib.sha256(txs[slot].encode()).hexdigest() == e['tx_sha256'], k
        assert (int(res['origin_idx'][slot]), int(res['mutant_idx'][slot])) == (e['origin_idx'], e['mutant_idx']), k
    for k in range(0, n, 157):                                 # 64 pairs against the oracle
        r = oracle.solve(origins[k], mutants[k], L=4, mode=1, alntype=1, diag_range=(-200, 200),
                         match=1, mismatch=-3, go=-5, ge=-2)
        assert (res['opt_i'][k], res['opt_j'][k]) == r['opt'] and res['score'][k] == r['score'], k
        assert txs[k] == r['transcript'], k
    for other in ('packed16_plain', 'packed16_unscaled', 'packed16_plain_unscaled', 'int32', 'f64'):
        _, res2, txs2 = runs[other]
        assert (res2 == res).all(), other
        assert txs2 == txs, other


def test_packed16_admission_with_mismatch_above_match(oracle):
    """A mismatch score above the match score bounds the running score by min(X, Y) * mismatch: a 7000 x 7000 local
    band (about 35 000 at the optimum) must not take the 16-bit kernel, and a 1000 x 1000 one (bounded by 7000) may and
    must still equal the 32-bit kernel and the oracle.  (Round 3: a mismatch score above 0 reaches the packed kernels
    through their matrix form only -- which needs a score <= 0 in the matrix, here the match score; the plain form would let
    cells outside the table creep up from the sentinel: DESIGN.md section 5.  With 1 / 6 the 32-bit kernels run.)"""
    from biseqt_amd import _pwlib as W
    from biseqt_amd import synth
    from biseqt_amd.batch import BatchAligner
    rng = synth.rng_for(242)
    kw = dict(alnmode=1, alntype=1, alphabet_len=4, diag_range=(-10, 10), match_score=-1, mismatch_score=7,
              go_score=-5, ge_score=-2)
    for n, packed in ((7000, False), (1000, True)):
        o = synth.rand_seqs(rng, 1, n)[0]
        m = synth.rand_seqs(rng, 1, n)[0]
        pairs = [(o, m)] * 300                              # enough pairs to leave latency mode
        with BatchAligner(pairs, **dict(kw, match_score=1, mismatch_score=6)) as b:
            assert 'k_fill16' not in b.kernel_name, b.kernel_name
        with BatchAligner(pairs, **kw) as b:
            assert ('k_fill16' in b.kernel_name) == packed and (not packed or 'matrix' in b.kernel_name), b.kernel_name
            res = b.run()
            txs = b.transcripts(res)
        with BatchAligner(pairs, flags=W.PW_FLAG_NO_PACKED16, **kw) as b:
            assert 'k_fill16' not in b.kernel_name
            res2 = b.run()
            txs2 = b.transcripts(res2)
        assert (res == res2).all() and txs == txs2
        r = oracle.solve(o, m, L=4, mode=1, alntype=1, diag_range=(-10, 10), match=-1, mismatch=7, go=-5, ge=-2)
        assert (res['opt_i'][0], res['opt_j'][0]) == r['opt'] and res['score'][0] == r['score']
        assert txs[0] == r['transcript']
        if not packed:
            assert r['score'] > 32767


def test_scaled_packed_kernel_at_its_admission_bound(oracle):
    """The packed kernel that holds every score times 4 (WaveFill16 RULE 3) is taken while min(X, Y) * max(match, mismatch)
    stays below 2048: identical sequences whose score reaches 2045 and 2047 must take it and equal the unscaled packed kernel,
    the 32-bit kernel and the oracle; one letter more (2050) must fall back to the unscaled kernel."""
    from biseqt_amd import _pwlib as W
    from biseqt_amd import synth
    from biseqt_amd.batch import BatchAligner
    rng = synth.rng_for(77)
    for n, match, mismatch, scaled in ((409, 5, -4, True), (2047, 1, -3, True), (2047, 1, 0, True), (410, 5, -4, False),
                                       (341, -2, 6, True), (342, -2, 6, False)):     # (mismatch above 0: the matrix form)
        o = synth.rand_seqs(rng, 1, n)[0]
        m2 = synth.mutate(rng, o, 0.05, 0.02, 0.3)
        kw = dict(alnmode=1, alntype=1, alphabet_len=4, diag_range=(-200, 200), match_score=match, mismatch_score=mismatch,
                  go_score=-5, ge_score=-2)
        pairs = [(o, o.copy())] * 150 + [(o, m2)] * 150 + [(m2, o)] * 20    # enough pairs to leave latency mode
        if min(len(o), len(m2)) < n and scaled is False:
            pairs = [(o, o.copy())] * 320
        runs = []
        for env, flags in (('', 0), ('1', 0), ('', W.PW_FLAG_NO_PACKED16)):
            if env:
                os.environ['PWLIB_NO_SCALED16'] = env
            try:
                with BatchAligner(pairs, flags=flags, **kw) as b:
                    name = b.kernel_name
                    res = b.run()
                    runs.append((name, res.copy(), b.transcripts(res)))
            finally:
                os.environ.pop('PWLIB_NO_SCALED16', None)
        assert ('x4' in runs[0][0]) == scaled, (n, match, runs[0][0])
        assert 'k_fill16' in runs[1][0] and 'x4' not in runs[1][0]
        assert 'k_fill16' not in runs[2][0]
        for other in runs[1:]:
            assert (other[1] == runs[0][1]).all() and other[2] == runs[0][2], (n, match, other[0])
        for k in (0, len(pairs) - 1):
            r = oracle.solve(pairs[k][0], pairs[k][1], L=4, mode=1, alntype=1, diag_range=(-200, 200), match=match,
                             mismatch=mismatch, go=-5, ge=-2)
            assert (runs[0][1]['opt_i'][k], runs[0][1]['opt_j'][k]) == r['opt'] and runs[0][1]['score'][k] == r['score']
            assert runs[0][2][k] == r['transcript']
        if scaled and mismatch < match:
            assert runs[0][1]['score'][0] == n * match


def test_negative_match_score_on_the_packed_kernels(oracle):
    """The API takes any scores (/root/reference/biseqt/pw.py:185-196 only builds the matrix): a NEGATIVE match score with
    a worse, an equal or a better mismatch score, on the packed kernels of every rule (B_LOCAL, B_OVERLAP, B_GLOBAL,
    standard-mode GLOBAL) -- each batch must equal the 32-bit kernel pair for pair and the oracle on a sample."""
    from biseqt_amd import _pwlib as W
    from biseqt_amd import synth
    from biseqt_amd.batch import BatchAligner
    rng = synth.rng_for(911)
    seen = set()
    for (match, mismatch, go, ge) in ((-1, -2, -3, -1), (-1, 2, -2, -1), (-2, 0, -1, 0), (-1, -1, 0, -1)):
        for mode, alntype, dr in ((1, 1, (-12, 9)), (1, 2, (-30, 25)), (1, 0, (-20, 20)), (0, 0, None)):
            pairs = []
            for _ in range(300):                                # enough pairs to leave latency mode
                n = int(rng.integers(20, 400 if mode else 180))
                o = synth.rand_seqs(rng, 1, n)[0]
                pairs.append((o, synth.mutate(rng, o, 0.06, 0.03, 0.4)))
            kw = dict(alnmode=mode, alntype=alntype, alphabet_len=4, match_score=match, mismatch_score=mismatch,
                      go_score=go, ge_score=ge, check_band=False)   # short pairs: the C side clamps the band like dptable_init
            if dr is not None:
                kw['diag_range'] = dr
            with BatchAligner(pairs, **kw) as b:
                name = b.kernel_name
                res = b.run().copy()
                txs = b.transcripts(res)
                rcs = [b.init_rc(k) for k in range(len(pairs))]
            with BatchAligner(pairs, flags=W.PW_FLAG_NO_PACKED16, **kw) as b:
                assert 'k_fill16' not in b.kernel_name
                res2 = b.run().copy()
                txs2 = b.transcripts(res2)
            assert 'k_fill16' in name, (name, kw)
            seen.add(name)
            assert (res == res2).all() and txs == txs2, (name, kw)
            for k in range(0, len(pairs), 23):
                r = oracle.solve(pairs[k][0], pairs[k][1], L=4, mode=mode, alntype=alntype, diag_range=dr, match=match,
                                 mismatch=mismatch, go=go, ge=ge)
                where = (k, name, kw)
                assert rcs[k] == r['init_rc'], where
                if r['init_rc'] != 0:
                    continue
                assert (int(res['opt_i'][k]), int(res['opt_j'][k])) == tuple(r['opt']), where
                if res['opt_i'][k] == -1:
                    continue
                st = int(res['status'][k])
                assert res['score'][k] == r['score'], where
                assert bool(st & W.PW_ST_PANICK) == bool(r['would_panick']), where
                if not (st & (W.PW_ST_PANICK | W.PW_ST_EMPTY)):
                    assert txs[k] == r['transcript'], where
                    assert (res['origin_idx'][k], res['mutant_idx'][k]) == (r['origin_idx'], r['mutant_idx']), where
    assert len(seen) >= 3, seen                                  # rules 0 (scaled), 1 and 2


def test_standard_mode_global_and_overlap_on_the_packed_kernels(oracle):
    """Standard-mode GLOBAL and OVERLAP batches take the packed rule-2 / rule-1 kernels -- the global / overlap rules on the
    band [-Y, X], OVERLAP with its own order of ties among the last cells -- and must equal the 32-bit kernel for every pair
    and the oracle on a sample; unrelated pairs drive the scores far below zero, suffix-prefix pairs make last cells tie,
    empty and one-letter sequences sit in the same batch."""
    from biseqt_amd import _pwlib as W
    from biseqt_amd import synth
    from biseqt_amd.batch import BatchAligner
    rng = synth.rng_for(99)
    pairs = []
    for k in range(1200):
        n = int(rng.integers(0, 600)) if k % 7 else int(rng.integers(0, 3))
        o = synth.rand_seqs(rng, 1, n)[0]
        if k % 3 == 0:
            m = synth.rand_seqs(rng, 1, int(rng.integers(0, 600)))[0]
        elif k % 3 == 1 and n > 20:                        # a suffix of o is a prefix of m
            m = np.concatenate([o[int(rng.integers(0, n)):], synth.rand_seqs(rng, 1, int(rng.integers(0, 80)))[0]])
        else:
            m = synth.mutate(rng, o, 0.06, 0.03, 0.4)
        pairs.append((o, m))
    for alntype, tag in ((0, ', 2>'), (4, ', 1>'), (5, ', 1>'), (6, ', 1>')):   # GLOBAL -> rule 2; OVERLAP, START_- and END_ANCHORED_OVERLAP -> rule 1
        kw = dict(alnmode=0, alntype=alntype, alphabet_len=4, match_score=2, mismatch_score=-3, go_score=-4, ge_score=-1)
        runs = []
        for flags in (0, W.PW_FLAG_NO_PACKED16):
            with BatchAligner(pairs, flags=flags, **kw) as b:
                name = b.kernel_name
                res = b.run()
                runs.append((name, res.copy(), b.transcripts(res)))
        assert 'k_fill16' in runs[0][0] and tag in runs[0][0], runs[0][0]
        assert 'k_fill16' not in runs[1][0]
        assert (runs[0][1] == runs[1][1]).all() and runs[0][2] == runs[1][2], alntype
        for k in range(0, 1200, 23):
            r = oracle.solve(pairs[k][0], pairs[k][1], L=4, mode=0, alntype=alntype, match=2, mismatch=-3, go=-4, ge=-1)
            assert (runs[0][1]['opt_i'][k], runs[0][1]['opt_j'][k]) == r['opt'] and runs[0][1]['score'][k] == r['score'], (alntype, k)
            if r['would_panick']:                          # (the reference exits here, pw.c:132-134: reported as a status)
                assert runs[0][1]['status'][k] & W.PW_ST_PANICK, (alntype, k)
            else:
                assert (runs[0][2][k] or '') == (r['transcript'] or ''), (alntype, k)


def test_packed_body_on_several_wavefronts_per_pair(oracle):
    """K2a with the 16-bit body (`k_fill16_mw`): batches of more than 256 pairs whose bands need 2049 .. 16 384 diagonals --
    standard-mode tables of 1 .. 4 kb, wide bands on longer pairs -- for the four packed rules: records and transcripts equal
    to the 32-bit kernels for every pair, a sample equal to the oracle (ragged lengths, unrelated pairs in the batch)."""
    from biseqt_amd import _pwlib as W
    from biseqt_amd import synth
    from biseqt_amd.batch import BatchAligner
    rng = synth.rng_for(191)
    cases = ((300, 1300, 0, 1, None, (1, -3, -5, -2), 'k_fill16_mw<8, 3>'), (300, 1100, 0, 0, None, (2, -3, -4, -1), 'k_fill16_mw<8, 2>'),
             (280, 1500, 0, 4, None, (1, -1, -1, -1), 'k_fill16_mw<8, 1>'), (260, 4000, 1, 1, (-1500, 1400), (1, -3, -5, -2), 'k_fill16_mw<8, 0>'),
             (260, 3000, 1, 2, (-2600, 2500), (1, -3, 0, -2), 'k_fill16_mw<12, 1>'))
    for count, n, mode, alntype, band, sc, kernel in cases:
        pairs = []
        for k in range(count):
            o = synth.rand_seqs(rng, 1, n if k % 7 else int(rng.integers(0, n)))[0]
            m = synth.rand_seqs(rng, 1, n if k % 5 else int(rng.integers(1, n)))[0] if k % 3 == 0 else synth.mutate(rng, o, 0.05, 0.02, 0.4)
            pairs.append((o, m))
        kw = dict(alnmode=mode, alntype=alntype, alphabet_len=4, match_score=sc[0], mismatch_score=sc[1], go_score=sc[2],
                  ge_score=sc[3], check_band=False)
        okw = dict(L=4, mode=mode, alntype=alntype, match=sc[0], mismatch=sc[1], go=sc[2], ge=sc[3])
        if band is not None:
            kw['diag_range'] = band; okw['diag_range'] = band
        runs = []
        for flags in (0, W.PW_FLAG_NO_PACKED16):
            with BatchAligner(pairs, flags=flags, **kw) as b:
                name = b.kernel_name
                res = b.run()
                runs.append((name, res.copy(), b.transcripts(res)))
        assert kernel in runs[0][0], (runs[0][0], kernel)
        assert 'k_fill16' not in runs[1][0]
        assert (runs[0][1] == runs[1][1]).all() and runs[0][2] == runs[1][2], kernel
        for k in range(0, count, 29):
            r = oracle.solve(pairs[k][0], pairs[k][1], **okw)
            if r['init_rc'] != 0:
                continue
            assert (runs[0][1]['opt_i'][k], runs[0][1]['opt_j'][k]) == r['opt'], (kernel, k)
            if r['opt'][0] >= 0:
                assert runs[0][1]['score'][k] == r['score'], (kernel, k)
                if not r['would_panick']:
                    assert (runs[0][2][k] or '') == (r['transcript'] or ''), (kernel, k)


def test_traceback_from_explicit_end_cells(oracle):
    """dptable_traceback accepts any end cell (pw.c:116-123), not only the optimum."""
    from biseqt_amd.batch import BatchAligner
    rng = np.random.default_rng(17)
    o, m = rng.integers(0, 4, 50), rng.integers(0, 4, 47)
    with BatchAligner([(o, m)], alnmode=0, alntype=0, alphabet_len=4, match_score=1, mismatch_score=-1,
                      go_score=-1, ge_score=-1) as b:
        b.solve()
        b.traceback_from([[30, 28]])
        b.sync()
        res = b.results()
        tx = b.transcripts(res)[0]
    r = oracle.solve(o[:30], m[:28], L=4, match=1, mismatch=-1, go=-1, ge=-1)   # global: prefix problem
    assert tx == r['transcript']


def test_wide_bands_multi_wavefront(oracle):
    """Bands wider than one wavefront holds (> 2048 diagonals): the multi-wavefront kernel (LDS halos between
    the wavefronts of a workgroup), all score types / variants, through the batch API and through Aligner."""
    from biseqt_amd import synth, _pwlib as W
    from biseqt_amd.batch import BatchAligner
    rng = synth.rng_for(31)
    o = synth.rand_seqs(rng, 1, 1500)[0]
    m = synth.mutate(rng, o, 0.08, 0.04, 0.4)
    o2 = synth.rand_seqs(rng, 1, 2600)[0]
    m2 = synth.mutate(rng, o2[300:2400], 0.05, 0.03, 0.3)
    cases = [
        (o, m, dict(mode=0, alntype=0), 0),                                   # STD GLOBAL, ~3000 diagonals
        (o, m, dict(mode=0, alntype=1), 0),                                   # STD LOCAL
        (o, m, dict(mode=0, alntype=4), W.PW_FLAG_FORCE_F64),                 # STD OVERLAP, f64
        (o2, m2, dict(mode=1, alntype=2, diag_range=(-1900, 2500)), 0),       # B_OVERLAP, 4401 diagonals: 3 wavefronts
        (o2, m2, dict(mode=1, alntype=1, diag_range=(-1900, 2500)), W.PW_FLAG_FORCE_GENERIC),
    ]
    for o_, m_, kw, flags in cases:
        bkw = dict(alnmode=kw['mode'], alntype=kw['alntype'], alphabet_len=4, match_score=1, mismatch_score=-3,
                   go_score=-5, ge_score=-2, flags=flags)
        if 'diag_range' in kw:
            bkw['diag_range'] = kw['diag_range']
        with BatchAligner([(o_, m_), (m_, o_)] if kw['mode'] == 0 else [(o_, m_)], **bkw) as b:
            assert 'k_fill' in b.kernel_name
            res = b.run()
            txs = b.transcripts(res)
        r = oracle.solve(o_, m_, L=4, match=1, mismatch=-3, go=-5, ge=-2, **kw)
        assert (res['opt_i'][0], res['opt_j'][0]) == r['opt'], kw
        assert res['score'][0] == r['score'] and txs[0] == r['transcript'], kw
        assert (res['origin_idx'][0], res['mutant_idx'][0]) == (r['origin_idx'], r['mutant_idx'])
        if kw['mode'] == 0:
            r2 = oracle.solve(m_, o_, L=4, match=1, mismatch=-3, go=-5, ge=-2, **kw)
            assert res['score'][1] == r2['score'] and txs[1] == r2['transcript'], kw


def test_aligner_standard_mode_two_kb(oracle):
    """A 2 kb x 2 kb standard-mode problem through the drop-in ABI (4001 diagonals: two wavefronts)."""
    from biseqt_amd import synth
    from biseqt_amd.pw import Aligner, LOCAL
    from biseqt_amd.sequence import Alphabet, Sequence
    rng = synth.rng_for(8)
    o = synth.rand_seqs(rng, 1, 2000)[0]
    m = synth.mutate(rng, o, 0.1, 0.05, 0.3)
    A = Alphabet('ACGT')
    with Aligner(Sequence(A, o.tolist()), Sequence(A, m.tolist()), alntype=LOCAL, match_score=1,
                 mismatch_score=-3, go_score=-5, ge_score=-2) as aligner:
        score = aligner.solve()
        aln = aligner.traceback()
        table = aligner.table_scores()
    r = oracle.solve(o, m, L=4, alntype=oracle.LOCAL, match=1, mismatch=-3, go=-5, ge=-2, want_table=True)
    assert score == r['score'] and aln.transcript == r['transcript']
    assert (aln.origin_start, aln.mutant_start) == (r['origin_idx'], r['mutant_idx'])
    H = r['H'].reshape(len(o) + 1, len(m) + 1)
    assert np.array_equal(np.array(table), H[:len(o), :len(m)])


@pytest.mark.parametrize('radius,n', [(4, 37), (16, 50), (50, 23), (120, 11)])
def test_lane_packed_batches(oracle, radius, n):
    """Narrow bands: several pairs share a wavefront (lane packing), including a ragged last wavefront and
    pairs of different lengths inside one wavefront (different block counts and steady ranges)."""
    from biseqt_amd import synth
    from biseqt_amd.batch import BatchAligner
    rng = synth.rng_for(100 + radius)
    pairs = []
    for k in range(n):
        o = synth.rand_seqs(rng, 1, int(rng.integers(40, 700)))[0]
        pairs.append((o, synth.mutate(rng, o, 0.06, 0.03, 0.3)))
    for typ, mode, dr in ((1, 1, (-radius, radius)),):
        with BatchAligner(pairs, alnmode=mode, alntype=typ, alphabet_len=4, diag_range=dr, check_band=False,
                          match_score=2, mismatch_score=-3, go_score=-4, ge_score=-1) as b:
            name = b.kernel_name
            res = b.run()
            txs = b.transcripts(res)
        assert 'k_fill16' in name, name
        assert ('true' in name) == (radius <= 50), name          # narrow bands: the lane-packed form is chosen
        for k, (o, m) in enumerate(pairs):
            r = oracle.solve(o, m, L=4, mode=mode, alntype=typ, diag_range=dr, match=2, mismatch=-3, go=-4, ge=-1)
            assert (res['opt_i'][k], res['opt_j'][k]) == r['opt'], (radius, k)
            assert res['score'][k] == r['score'] and txs[k] == r['transcript'], (radius, k)
            assert (res['origin_idx'][k], res['mutant_idx'][k]) == (r['origin_idx'], r['mutant_idx'])


def test_tiled_kernel_forced(oracle):
    """The time-blocked, ghost-zone tiled kernel (K2b, for tables wider than 16384 diagonals) forced on tables the
    oracle can check: several tiles x dozens of time blocks, all end rules, int32 / f64 / generic."""
    from biseqt_amd import synth, _pwlib as W
    from biseqt_amd.batch import BatchAligner
    rng = synth.rng_for(77)
    o = synth.rand_seqs(rng, 1, 1500)[0]
    m = synth.mutate(rng, o, 0.08, 0.04, 0.4)
    o2 = synth.rand_seqs(rng, 1, 900)[0]
    m2 = np.concatenate([synth.rand_seqs(rng, 1, 700)[0], synth.mutate(rng, o2[200:], 0.05, 0.03, 0.3)])
    cases = [
        (o, m, dict(mode=0, alntype=1), 0),                                  # STD LOCAL: 5 tiles
        (o, m, dict(mode=0, alntype=0), 0),                                  # STD GLOBAL
        (o2, m2, dict(mode=0, alntype=4), 0),                                # STD OVERLAP
        (o2, m2, dict(mode=0, alntype=2), W.PW_FLAG_FORCE_F64),              # START_ANCHORED, f64
        (o, m, dict(mode=1, alntype=1, diag_range=(-700, 900)), 0),          # B_LOCAL
        (o, m, dict(mode=1, alntype=2, diag_range=(-1200, 300)), W.PW_FLAG_FORCE_GENERIC),   # B_OVERLAP, generic
        (o[:37], m[:41], dict(mode=0, alntype=1), 0),                        # a single small tile
    ]
    for o_, m_, kw, flags in cases:
        bkw = dict(alnmode=kw['mode'], alntype=kw['alntype'], alphabet_len=4, match_score=1, mismatch_score=-3,
                   go_score=-5, ge_score=-2, flags=flags | W.PW_FLAG_FORCE_TILED)
        if 'diag_range' in kw:
            bkw['diag_range'] = kw['diag_range']
        with BatchAligner([(o_, m_), (m_[:300], o_[:400])] if kw['mode'] == 0 else [(o_, m_)], **bkw) as b:
            assert 'tile' in b.kernel_name
            res = b.run()
            txs = b.transcripts(res)
        r = oracle.solve(o_, m_, L=4, match=1, mismatch=-3, go=-5, ge=-2, **kw)
        assert (res['opt_i'][0], res['opt_j'][0]) == r['opt'], kw
        if r['opt'][0] != -1:
            assert res['score'][0] == r['score'], kw
            if not r['would_panick'] and not r['tb_null']:
                assert txs[0] == r['transcript'], kw
                assert (res['origin_idx'][0], res['mutant_idx'][0]) == (r['origin_idx'], r['mutant_idx'])
        if kw['mode'] == 0:
            r2 = oracle.solve(m_[:300], o_[:400], L=4, match=1, mismatch=-3, go=-5, ge=-2, **kw)
            assert (res['opt_i'][1], res['opt_j'][1]) == r2['opt'], kw
            if r2['opt'][0] != -1:
                assert res['score'][1] == r2['score'], kw
                if not r2['would_panick'] and not r2['tb_null']:
                    assert txs[1] == r2['transcript'], kw


def test_wide_table_kernels_natural_vs_oracle(oracle):
    """A table wider than one workgroup holds (9000 x ~9000 standard mode, 18 000 diagonals) goes by itself to the
    strip pipeline (K2c: integer scores, simple scoring) or to the tiled kernel (K2b: f64 here); the oracle still
    fits (8e7 cells): score, end cell, start and transcript must be identical.  The wide pair shares its batch with
    small pairs that take the ordinary one-wavefront kernel."""
    from biseqt_amd import synth, _pwlib as W
    from biseqt_amd.batch import BatchAligner
    rng = synth.rng_for(31)
    o = synth.rand_seqs(rng, 1, 9000)[0]
    m = synth.mutate(rng, o, 0.07, 0.015, 0.5)
    small = [(o[:300], m[:310]), (m[:90], o[:70])]
    for alntype, flags, kernel in ((1, 0, 'strip'), (4, 0, 'strip'), (0, 0, 'strip'), (1, W.PW_FLAG_FORCE_F64, 'tile')):
        with BatchAligner([small[0], (o, m), small[1]], alnmode=0, alntype=alntype, alphabet_len=4, match_score=1,
                          mismatch_score=-3, go_score=-5, ge_score=-2, flags=flags) as b:
            res = b.run()
            txs = b.transcripts(res)
        with BatchAligner([(o, m)], alnmode=0, alntype=alntype, alphabet_len=4, match_score=1,
                          mismatch_score=-3, go_score=-5, ge_score=-2, flags=flags) as b:
            assert kernel in b.kernel_name, b.kernel_name
        for k, (oo, mm) in enumerate([small[0], (o, m), small[1]]):
            r = oracle.solve(oo, mm, L=4, mode=0, alntype=alntype, match=1, mismatch=-3, go=-5, ge=-2)
            assert (res['opt_i'][k], res['opt_j'][k]) == r['opt'], (alntype, k)
            if r['opt'][0] != -1:
                assert res['score'][k] == r['score'] and txs[k] == r['transcript'], (alntype, k)
                assert (res['origin_idx'][k], res['mutant_idx'][k]) == (r['origin_idx'], r['mutant_idx'])


def test_strip_pipeline_forced_on_small_tables(oracle):
    """The strip pipeline (K2c, pw_strip.h) forced on standard-mode tables the oracle can check: all seven alignment
    types, seven score sets (linear and affine gaps, mismatch above match), 1 - 3 pairs per batch running one after
    another, every batch solved twice (FIFO granules of the first solve must never pass for fresh ones), tables from
    empty to 40 strips."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('strip_check', os.path.join(os.path.dirname(__file__), 'micro', 'strip_check.py'))
    sc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sc)
    assert sc.run(cases=120, seed=20261004, maxlen=900) == 0
    assert sc.run(cases=20, seed=20261005, maxlen=2600) == 0
    # round 3: a substitution matrix over 4 letters whose entries fit a signed byte runs on the strips' byte rows too
    # (_alnchoice_M reads subst_scores[o][m], _pw_internals.c:217-245; before, such pairs went to the tiled kernel)
    assert sc.run(cases=70, seed=20261006, maxlen=900, matrices=True) == 0


def test_config3_full_size_properties():
    """BASELINE config 3: ONE 100 kb x ~100 kb pair, standard mode LOCAL, 1/-3/-5/-2 (1.0e10 cells, 200 k diagonals;
    the reference would need ~0.5 TB, so no oracle exists -- SURVEY 8c).  Size-independent check: re-scoring the
    transcript with the reference's rule reproduces the reported score, the path runs from the reported start to
    the reported end cell, every M / S agrees with the letters; and the score is at least that of the best
    alignment of a 6 kb window the oracle can solve."""
    from biseqt_amd import synth
    from biseqt_amd.batch import BatchAligner
    rng = synth.rng_for(3)
    o = synth.rand_seqs(rng, 1, 100000)[0]
    m = synth.mutate(rng, o, 0.07, 0.015, 0.5)
    with BatchAligner([(o, m)], alnmode=0, alntype=1, alphabet_len=4, match_score=1, mismatch_score=-3,
                      go_score=-5, ge_score=-2) as b:
        assert 'strip' in b.kernel_name and b.cells == (len(o) + 1) * (len(m) + 1)
        res = b.run()
        tx = b.transcripts(res)[0]
    s, i, j = _rescore(o, m, tx, int(res['origin_idx'][0]), int(res['mutant_idx'][0]), 1, -3, -5, -2)
    assert s == res['score'][0] and s > 30000
    assert (i, j) == (int(res['opt_i'][0]), int(res['opt_j'][0]))
    assert tx[0] in 'MS' and tx[-1] in 'MS'                  # a local alignment never starts or ends with a gap
    # two independent kernels, one answer: the time-blocked tiled kernel (diagonal lanes, ghost zones) at full size
    from biseqt_amd import _pwlib as W
    with BatchAligner([(o, m)], alnmode=0, alntype=1, alphabet_len=4, match_score=1, mismatch_score=-3,
                      go_score=-5, ge_score=-2, flags=W.PW_FLAG_FORCE_TILED) as b:
        assert 'tile' in b.kernel_name
        res2 = b.run()
        tx2 = b.transcripts(res2)[0]
    assert (res2 == res).all() and tx2 == tx


def test_band_edge_never_leaks_long_pairs(oracle):
    """The true alignment lies exactly on the first diagonal OUTSIDE the band (origin == mutant, band (-21, -1)):
    nothing of that diagonal's ~9000-long perfect run may leak into the band.  (The 16-bit kernel's band-edge
    block is only 8192 deep, so pairs this long must take the 32-bit kernel.)"""
    from biseqt_amd import synth
    from biseqt_amd.batch import BatchAligner
    rng = synth.rng_for(91)
    o = synth.rand_seqs(rng, 1, 9000)[0]
    for band in ((-21, -1), (1, 23)):
        with BatchAligner([(o, o.copy())] * 3, alnmode=1, alntype=1, alphabet_len=4, diag_range=band,
                          match_score=1, mismatch_score=-3, go_score=-5, ge_score=-2) as b:
            res = b.run()
            txs = b.transcripts(res)
        r = oracle.solve(o, o, L=4, mode=1, alntype=1, diag_range=band, match=1, mismatch=-3, go=-5, ge=-2)
        for k in range(3):
            assert res['score'][k] == r['score'] and (res['opt_i'][k], res['opt_j'][k]) == r['opt'], (band, res[k], r['score'])
            assert txs[k] == r['transcript']


def test_adversarial_fuzz_vs_oracle(oracle):
    """A bounded run of tests/micro/fuzz_gpu.py: random batches over all modes / types / score sets / kernel-forcing
    flags with adversarial pair shapes (alignments on and next to band edges, identical and shifted copies,
    repeats, empty sequences, clamped and infeasible bands), every pair compared with the oracle."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('fuzz_gpu', os.path.join(os.path.dirname(__file__), 'micro', 'fuzz_gpu.py'))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    nb, npairs, nbad = fz.run(60, 20261004, max_batches=400)
    assert nbad == 0 and npairs > 1000
    nb, npairs, nbad = fz.run(60, 20261005, long_mode=True, max_batches=25)
    assert nbad == 0
    # (round 3) few pairs, bands thousands of diagonals wide and far off the main diagonal: tiles, strips, multi-wavefront
    # kernels on tables whose diagonals start and end at very different steps -- the regime of the round's late find
    nb, npairs, nbad = fz.run(40, 20261006, long_mode='wide', max_batches=120)
    assert nbad == 0 and npairs > 100


@pytest.mark.parametrize('latency_mode', ['0', '1'])
def test_wide_pairs_single_wavefront_and_latency_mode(oracle, latency_mode, monkeypatch):
    """Bands of 300 .. 2000 diagonals both ways: one wavefront per pair with 8 .. 32 diagonals per lane
    (PWLIB_LATENCY_MODE=0, the throughput layout) and spread over up to 8 wavefronts with 4 .. 16 per lane
    (PWLIB_LATENCY_MODE=1, what small batches get by default) -- all types, int32 / f64 / generic."""
    from biseqt_amd import synth, _pwlib as W
    from biseqt_amd.batch import BatchAligner
    monkeypatch.setenv('PWLIB_LATENCY_MODE', latency_mode)
    rng = synth.rng_for(404)
    cases = [(700, 1, 1, (-150, 160), 0), (900, 1, 2, (-300, 280), 0), (1000, 1, 0, (-500, 600), W.PW_FLAG_FORCE_F64),
             (800, 0, 1, None, 0), (1000, 0, 0, None, W.PW_FLAG_FORCE_GENERIC), (600, 0, 4, None, 0), (950, 0, 6, None, 0)]
    for n, mode, alntype, band, flags in cases:
        o = synth.rand_seqs(rng, 1, n)[0]
        m = synth.mutate(rng, o, .08, .03, .4)
        kw = dict(alnmode=mode, alntype=alntype, alphabet_len=4, match_score=1, mismatch_score=-3, go_score=-5, ge_score=-2,
                  flags=flags)
        okw = dict(L=4, mode=mode, alntype=alntype, match=1, mismatch=-3, go=-5, ge=-2)
        if band is not None:
            kw['diag_range'] = band; okw['diag_range'] = band
        with BatchAligner([(o, m), (m, o)] if band is None else [(o, m)], **kw) as b:
            name = b.kernel_name
            res = b.run()
            txs = b.transcripts(res)
        ndiag = (band[1] - band[0] + 1) if band is not None else len(o) + len(m) + 1
        strip_ok = mode == 0 and not flags            # standard mode, integer scores, simple scoring: the strip pipeline
        if latency_mode == '0':
            # (f64 with 32 diagonals per lane does not fit one wavefront's registers: several wavefronts in any mode)
            assert ('k_fill_mw' not in name or (flags & W.PW_FLAG_FORCE_F64 and ndiag > 1024)) and 'strip' not in name, name
        elif strip_ok:
            # a couple of pairs: row strips, one pair after another -- or the 16-bit body on several wavefronts per pair
            assert 'k_fill_strip' in name or 'k_fill16_mw' in name, (name, ndiag)
        elif ndiag > 1024:                       # (up to 64 x 12 diagonals the packed one-wavefront kernel may still win)
            # several wavefronts (32-bit or 16-bit body), or the tiled kernel for one pair
            assert 'k_fill_mw' in name or 'k_fill16_mw' in name or 'tile' in name, (name, ndiag)
        r = oracle.solve(o, m, **okw)
        assert (res['opt_i'][0], res['opt_j'][0]) == r['opt'], (n, mode, alntype, name)
        if r['opt'][0] != -1:
            assert res['score'][0] == r['score']
            if not r['would_panick'] and not r['tb_null']:
                assert txs[0] == r['transcript']
                assert (res['origin_idx'][0], res['mutant_idx'][0]) == (r['origin_idx'], r['mutant_idx'])


def test_packed_overlap_and_global_kernels_at_their_bounds(oracle):
    """The 16-bit kernels for B_OVERLAP / B_GLOBAL (WaveFill16<.., RULE = 1 | 2>) where their arithmetic is tightest:
    unrelated 4900-base pairs (scores near -15000, the sentinel is -24000), identical pairs with match 3 (+14700),
    the true alignment lying on the first diagonal OUTSIDE the band (its offer must be clamped away), bands whose
    width is not a multiple of the lane width, lane-packed and one-pair-per-wave layouts."""
    from biseqt_amd import synth
    from biseqt_amd.batch import BatchAligner
    rng = synth.rng_for(2026)
    a = synth.rand_seqs(rng, 1, 4900)[0]
    b_ = synth.rand_seqs(rng, 1, 4900)[0]
    near = synth.mutate(rng, a, .03, .01, .3)
    cases = []
    for alntype in (2, 0):                                   # B_OVERLAP, B_GLOBAL
        cases += [
            (a, b_, alntype, (-21, 20), dict(match_score=1, mismatch_score=-3, go_score=-5, ge_score=-2)),      # unrelated
            (a, a.copy(), alntype, (-7, 9), dict(match_score=3, mismatch_score=-3, go_score=-5, ge_score=-2)),   # +14700
            (a, a.copy(), alntype, (-21, -1) if alntype == 2 else (-21, 0), dict(match_score=1, mismatch_score=-3, go_score=-5, ge_score=-2)),
            (a, a.copy(), alntype, (1, 23) if alntype == 2 else (0, 23), dict(match_score=1, mismatch_score=-3, go_score=-5, ge_score=-2)),
            (a, near, alntype, (min(0, len(a) - len(near)) - 37, max(0, len(a) - len(near)) + 41), dict(match_score=2, mismatch_score=-1, go_score=0, ge_score=-1)),
            (a, b_, alntype, (-30, 33), dict(match_score=1, mismatch_score=-4, go_score=-9, ge_score=-3)),       # down to about -19700
            (a, a.copy(), alntype, (-5, 4), dict(match_score=6, mismatch_score=-4, go_score=-9, ge_score=-3)),  # up to +29400
        ]
    for o, m, alntype, band, sc in cases:
        for n in (1, 300):                                   # latency layout / throughput layout (lane packing)
            with BatchAligner([(o, m)] * n, alnmode=1, alntype=alntype, alphabet_len=4, diag_range=band, **sc) as bt:
                name = bt.kernel_name
                res = bt.run()
                txs = bt.transcripts(res)
            assert 'k_fill16' in name and name.endswith(', %d>' % (1 if alntype == 2 else 2)), name
            r = oracle.solve(o, m, L=4, mode=1, alntype=alntype, diag_range=band, match=sc['match_score'],
                             mismatch=sc['mismatch_score'], go=sc['go_score'], ge=sc['ge_score'])
            for k in (0, n - 1):
                assert (res['opt_i'][k], res['opt_j'][k]) == r['opt'], (alntype, band, name)
                assert res['score'][k] == r['score'], (alntype, band, name, res['score'][k], r['score'])
                if not r['would_panick'] and not r['tb_null']:
                    assert txs[k] == r['transcript']
                    assert (res['origin_idx'][k], res['mutant_idx'][k]) == (r['origin_idx'], r['mutant_idx'])


def test_packed_overlap_global_sweep_of_band_ends(oracle):
    """Regression family of tests/golden/packed_overlap_regression.json on the GPU: B_OVERLAP / B_GLOBAL with every band
    end in a range, so that the step of the earliest-ending diagonal takes every residue modulo the 16-step block."""
    import json
    from biseqt_amd.batch import BatchAligner
    rec = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'packed_overlap_regression.json')))
    o, m = np.array(rec['origin'], np.uint8), np.array(rec['mutant'], np.uint8)
    for alntype in (2, 0):
        bands = [(-84, hi) for hi in range(70, 104)] + [(lo, 60) for lo in range(-100, -70)]
        if alntype == 0:
            bands = [b for b in bands if b[0] <= len(o) - len(m) <= b[1]]
        for n in (1, 300):
            for chunk in range(0, len(bands), 8):
                bs = bands[chunk:chunk + 8]
                with BatchAligner([(o, m)] * (len(bs) * (n // 8 if n > 1 else 1)), alnmode=1, alntype=alntype, alphabet_len=4,
                                  diag_range=bs * (n // 8 if n > 1 else 1), match_score=1, mismatch_score=-3, go_score=-5,
                                  ge_score=-2) as bt:
                    assert 'k_fill16' in bt.kernel_name
                    res = bt.run()
                    txs = bt.transcripts(res)
                for k, band in enumerate(bs):
                    r = oracle.solve(o, m, L=4, mode=1, alntype=alntype, diag_range=band, match=1, mismatch=-3, go=-5, ge=-2)
                    assert (res['opt_i'][k], res['opt_j'][k]) == r['opt'] and res['score'][k] == r['score'], (alntype, band, n)
                    if not r['would_panick'] and not r['tb_null']:
                        assert txs[k] == r['transcript'], (alntype, band, n)


def test_drop_in_calls_from_several_threads(oracle):
    """Distinct dptables are independent (the reference has no shared mutable state; cffi / ctypes release the
    GIL around the calls): four threads run init / solve / traceback / free cycles concurrently."""
    import threading
    from biseqt_amd import synth, _pwlib as W
    from oracle import ref_driver as R
    lib = R.load(W.PWLIB_SO)
    origins, mutants = synth.pair_batch(21, 8, 600)
    probs, want = [], []
    for k in range(8):
        kw = dict(mode=1, alntype=k % 3, diag_range=(-60, 70), L=4, match=1., mismatch=-3., go=-5., ge=-2.)
        probs.append(lambda k=k, kw=kw: R.Problem(origins[k].tolist(), mutants[k].tolist(), **kw))
        want.append(oracle.solve(origins[k], mutants[k], **kw))
    errors = []

    def worker(tid):
        try:
            for it in range(40):
                k = (tid * 3 + it) % 8
                out = R.run(lib, probs[k]())
                w = want[k]
                assert out['opt'] == w['opt'] and out['score'] == w['score'] and out['transcript'] == w['transcript'], (tid, it, k)
        except Exception as e:          # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]


def test_buffer_pool_reuse_and_trim():
    """Device buffers of a destroyed batch are handed to the next batch of similar size (no growth over repeated
    create / destroy cycles) and pw_pool_trim gives them back."""
    import ctypes
    from biseqt_amd import synth, _pwlib as W
    from biseqt_amd.batch import BatchAligner
    lib = W.load()
    origins, mutants = synth.pair_batch(5, 600, 1500)
    kw = dict(alnmode=1, alntype=1, alphabet_len=4, diag_range=(-150, 150), match_score=1, mismatch_score=-3, go_score=-5, ge_score=-2)

    def used():
        free, total = ctypes.c_uint64(), ctypes.c_uint64()
        assert lib.pw_device_memory(0, ctypes.byref(free), ctypes.byref(total)) == 0
        return (total.value - free.value) / 2.0 ** 20

    with BatchAligner(list(zip(origins, mutants)), **kw) as b:      # warm-up: code objects, runtime scratch
        b.run()
    lib.pw_pool_trim()
    base = used()
    marks = []
    for it in range(6):
        with BatchAligner(list(zip(origins, mutants)), **kw) as b:
            res = b.run()
            assert (res['opt_i'] >= 0).all()
        marks.append(used())
    assert max(marks[1:]) - marks[1] < 32, marks          # steady after the first cycle
    assert marks[1] - base > 50                           # ... because the buffers are parked in the pool
    lib.pw_pool_trim()
    assert used() - base < 32


def test_library_loaded_before_torch_still_sees_the_device():
    """One HIP runtime per process whatever the import order: a fresh interpreter loads pwlib.so FIRST, solves a batch,
    then imports torch, which must find the same device, and the library must keep working afterwards."""
    import subprocess
    import sys
    code = r'''
import sys
sys.path.insert(0, %r)
import numpy as np
from biseqt_amd import _pwlib as W
from biseqt_amd.batch import BatchAligner
lib = W.load()
assert 'torch' not in sys.modules
assert lib.pw_device_count() >= 1
pairs = [(np.array([0, 1, 2, 3, 0, 1], np.uint8), np.array([0, 1, 3, 3, 0, 1], np.uint8))] * 4
def solve():
    with BatchAligner(pairs, alnmode=0, alntype=0, alphabet_len=4, match_score=1, mismatch_score=-1, go_score=0, ge_score=-1) as b:
        return b.run()['score'].tolist()
first = solve()
import torch
assert torch.cuda.is_available() and torch.cuda.device_count() >= 1
t = torch.arange(8, device='cuda').sum().item()
assert t == 28
assert solve() == first
maps = open('/proc/self/maps').read()
libs = sorted({l.split()[-1] for l in maps.splitlines() if 'libamdhip64' in l})
assert len(libs) == 1, libs
print('ok', first, libs)
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k != 'PWLIB_HIP_RUNTIME'}
    p = subprocess.run([sys.executable, '-c', code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       universal_newlines=True, timeout=600)
    assert p.returncode == 0 and p.stdout.startswith('ok'), (p.stdout[-500:], p.stderr[-1500:])
