"""The scoring surface of the HIP path off the headline fast path (round 3): score sets that used to fall onto the f64 or
the generic kernels -- dyadic-rational scores (exact scaling onto the integer kernels), small integer substitution
matrices (packed 16-bit kernel with a matrix), START_ANCHORED / END_ANCHORED (their own packed instantiations) -- must
report the fast kernels and stay bit-identical to the forced f64 / generic kernels and to the oracle.

Reference semantics: biseqt/pwlib/_pw_internals.c:161-299 (move generators), :303-414 (end cells); the reference's
arithmetic is IEEE double (pwlib.h:71-78, 141-152)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KEYS = ('init_rc', 'opt', 'score', 'transcript', 'origin_idx', 'mutant_idx', 'tb_null', 'would_panick')


def _pairs(rng, n, lo, hi, L=4, unrelated_every=5):
    from biseqt_amd import synth
    pairs = []
    for k in range(n):
        X = int(rng.integers(lo, hi))
        o = rng.integers(0, L, X).astype(np.uint8)
        if unrelated_every and k % unrelated_every == 0:
            m = rng.integers(0, L, int(rng.integers(lo, hi))).astype(np.uint8)
        elif k % 7 == 3 and X > 20:                        # a suffix of o is a prefix of m
            m = np.concatenate([o[int(rng.integers(0, X)):], rng.integers(0, L, int(rng.integers(0, 60))).astype(np.uint8)])
        else:
            m = synth.mutate(rng, o, 0.06, 0.03, 0.4, L)
        pairs.append((o, m))
    return pairs


def _run(pairs, flags=0, **kw):
    from biseqt_amd.batch import BatchAligner
    with BatchAligner(pairs, flags=flags, check_band=False, **kw) as b:
        name, dtype = b.kernel_name, b.score_dtype
        res = b.run().copy()
        txs = b.transcripts(res)
        rcs = [b.init_rc(k) for k in range(len(pairs))]
    return name, dtype, res, txs, rcs


def _check_vs_oracle(oracle, pairs, res, txs, rcs, okw, every, where):
    from biseqt_amd import _pwlib as W
    for k in range(0, len(pairs), every):
        r = oracle.solve(pairs[k][0], pairs[k][1], **okw)
        w = (where, k)
        assert rcs[k] == r['init_rc'], w
        if r['init_rc'] != 0:
            continue
        assert (int(res['opt_i'][k]), int(res['opt_j'][k])) == tuple(r['opt']), w
        if res['opt_i'][k] == -1:
            continue
        st = int(res['status'][k])
        assert res['score'][k] == r['score'], (w, res['score'][k], r['score'])
        assert bool(st & W.PW_ST_PANICK) == bool(r['would_panick']), w
        if not (st & (W.PW_ST_PANICK | W.PW_ST_EMPTY)):
            assert txs[k] == r['transcript'], w
            assert (res['origin_idx'][k], res['mutant_idx'][k]) == (r['origin_idx'], r['mutant_idx']), w


DYADIC_SETS = [(0.25, -1., 0., -1.),          # config 5's extension scores: match = 1 / p_min - 1 at p_min = 0.8 (pipeline.py)
               (0.5, -1.5, -2.5, -1.),
               (1.25, -0.75, -0.5, -0.25),
               (2., -3.125, -4.5, -0.0625),
               (0.001953125, -0.00390625, 0., -0.0009765625)]   # 2^-9 .. 2^-10: the deepest shift


@pytest.mark.parametrize('mode,alntype,dr', [(1, 1, (-40, 35)), (1, 2, (-50, 50)), (1, 0, (-60, 60)), (0, 1, None),
                                             (0, 0, None), (0, 4, None), (0, 2, None), (0, 3, None)],
                         ids=['B_LOCAL', 'B_OVERLAP', 'B_GLOBAL', 'LOCAL', 'GLOBAL', 'OVERLAP', 'START_ANCHORED', 'END_ANCHORED'])
def test_dyadic_scores_run_on_the_integer_kernels(oracle, mode, alntype, dr):
    """Scores that are multiples of 2^-k are scaled by 2^k onto the integer kernels (exact: every partial sum of dyadic
    rationals is exact in the reference's doubles): the batch must report an integer kernel, and its records and
    transcripts must equal the forced f64 kernel's for every pair and the oracle's on a sample."""
    from biseqt_amd import _pwlib as W
    from biseqt_amd import synth
    rng = synth.rng_for(3100 + 10 * mode + alntype)
    for si, (match, mismatch, go, ge) in enumerate(DYADIC_SETS):
        pairs = _pairs(rng, 320, 0 if si == 1 else 20, 400 if mode else 200)
        kw = dict(alnmode=mode, alntype=alntype, alphabet_len=4, match_score=match, mismatch_score=mismatch,
                  go_score=go, ge_score=ge)
        if dr is not None:
            kw['diag_range'] = dr
        name, dtype, res, txs, rcs = _run(pairs, **kw)
        name64, dtype64, res64, txs64, _ = _run(pairs, flags=W.PW_FLAG_FORCE_F64, **kw)
        assert dtype == 'i32' and 'double' not in name, (name, kw)
        assert dtype64 == 'f64' and 'double' in name64, name64
        assert (res == res64).all() and txs == txs64, (name, kw)
        okw = dict(L=4, mode=mode, alntype=alntype, diag_range=dr, match=match, mismatch=mismatch, go=go, ge=ge)
        _check_vs_oracle(oracle, pairs, res, txs, rcs, okw, 13, (name, kw))
        if si == 0 and (mode, alntype) in ((1, 1), (1, 2), (1, 0), (0, 1), (0, 0), (0, 4)):
            assert 'k_fill16' in name, name          # small scaled scores: the packed kernel


def test_dyadic_substitution_matrix_and_positive_gap_open(oracle):
    """Dyadic scaling also serves the generic path (a substitution matrix, go > 0): integer arithmetic, equal to f64."""
    from biseqt_amd import _pwlib as W
    from biseqt_amd import synth
    rng = synth.rng_for(3177)
    subst = [[1.5, -0.5, -2.25, -0.5], [-0.5, 1.25, -0.5, -2.], [-2.25, -0.5, 1.75, -0.75], [-0.5, -2., -0.75, 1.]]
    for mode, alntype, dr, go in ((1, 1, (-30, 30), -1.5), (0, 0, None, 0.75), (1, 2, (-45, 45), 0.25)):
        pairs = _pairs(rng, 300, 10, 250)
        kw = dict(alnmode=mode, alntype=alntype, alphabet_len=4, subst_scores=subst, go_score=go, ge_score=-0.75)
        if dr is not None:
            kw['diag_range'] = dr
        name, dtype, res, txs, rcs = _run(pairs, **kw)
        name64, dtype64, res64, txs64, _ = _run(pairs, flags=W.PW_FLAG_FORCE_F64, **kw)
        assert dtype == 'i32' and dtype64 == 'f64', (name, name64)
        assert (res == res64).all() and txs == txs64, (name, kw)
        okw = dict(L=4, mode=mode, alntype=alntype, diag_range=dr, subst=subst, go=go, ge=-0.75)
        _check_vs_oracle(oracle, pairs, res, txs, rcs, okw, 11, (name, kw))


def test_non_dyadic_scores_stay_on_f64(oracle):
    """0.1 is not a dyadic rational: such scores keep the f64 kernel (and PWLIB_NO_DYADIC=1 keeps dyadic ones there too)."""
    import os
    from biseqt_amd import synth
    rng = synth.rng_for(3199)
    pairs = _pairs(rng, 40, 20, 200)
    kw = dict(alnmode=1, alntype=1, alphabet_len=4, diag_range=(-30, 30), mismatch_score=-1., go_score=0., ge_score=-1.)
    name, dtype, _, _, _ = _run(pairs, match_score=0.1, **kw)
    assert dtype == 'f64', name
    os.environ['PWLIB_NO_DYADIC'] = '1'
    try:
        name, dtype, res, txs, rcs = _run(pairs, match_score=0.25, **kw)
    finally:
        os.environ.pop('PWLIB_NO_DYADIC', None)
    assert dtype == 'f64', name
    name2, dtype2, res2, txs2, _ = _run(pairs, match_score=0.25, **kw)
    assert dtype2 == 'i32' and (res == res2).all() and txs == txs2


def test_dyadic_drop_in_table_scores(oracle):
    """The drop-in's materialised table (Aligner.table_scores, pw.py:278-285) is scaled back too."""
    from biseqt_amd import pw
    from biseqt_amd.sequence import Alphabet
    from biseqt_amd import synth
    rng = synth.rng_for(3201)
    A = Alphabet('ACGT')
    o = rng.integers(0, 4, 60)
    m = synth.mutate(rng, o.astype(np.uint8), 0.1, 0.05, 0.3)
    S, T = A.parse(''.join('ACGT'[i] for i in o)), A.parse(''.join('ACGT'[i] for i in m))
    with pw.Aligner(S, T, alntype=pw.LOCAL, match_score=0.75, mismatch_score=-1.25, go_score=-0.5, ge_score=-0.25) as a:
        score = a.solve()
        table = np.array(a.table_scores())
        aln = a.traceback()
    r = oracle.solve(o, m, L=4, mode=0, alntype=1, match=0.75, mismatch=-1.25, go=-0.5, ge=-0.25, want_table=True)
    assert score == r['score'] and aln.transcript == r['transcript']
    H = r['H'].reshape(len(o) + 1, len(m) + 1)
    assert (table == H[:len(o), :len(m)]).all()


def test_abandoned_strip_pipeline_is_solved_again(oracle, monkeypatch):
    """A strip pipeline (K2c) whose waits run out of patience (PWLIB_STRIP_SPIN_LIMIT=1: the first poll that finds no granule
    gives up) must drain, and `pw_batch_results` must solve the pair again without the strips: the caller sees the right
    records and transcripts, never "no alignment".  A following solve with the default limit goes through the strips again.
    PWLIB_NO_STRIP_REPAIR=1 surfaces the abandonment as an error instead."""
    from biseqt_amd import _pwlib as W
    from biseqt_amd import synth
    from biseqt_amd.batch import BatchAligner
    rng = synth.rng_for(3301)
    pairs = []
    for n in (1500, 700, 2100):
        o = synth.rand_seqs(rng, 1, n)[0]
        pairs.append((o, synth.mutate(rng, o, 0.08, 0.04, 0.3)))
    for alntype in (1, 0, 4):                               # LOCAL, GLOBAL, OVERLAP
        kw = dict(alnmode=0, alntype=alntype, alphabet_len=4, match_score=1, mismatch_score=-3, go_score=-5, ge_score=-2)
        exp = [oracle.solve(o, m, L=4, mode=0, alntype=alntype, match=1, mismatch=-3, go=-5, ge=-2) for o, m in pairs]

        def check(res, txs):
            for k, r in enumerate(exp):
                assert (int(res['opt_i'][k]), int(res['opt_j'][k])) == tuple(r['opt']), (alntype, k)
                assert res['score'][k] == r['score'] and txs[k] == r['transcript'], (alntype, k)
                assert not (int(res['status'][k]) & W.PW_ST_BADPATH)

        with BatchAligner(pairs, flags=W.PW_FLAG_FORCE_STRIP, **kw) as b:
            assert 'k_fill_strip' in b.kernel_name
            monkeypatch.setenv('PWLIB_STRIP_SPIN_LIMIT', '1')
            res = b.run()
            check(res, b.transcripts(res))
            # explicit end cells after a repair: served by the replacement too
            b.traceback_from([(int(res['opt_i'][k]), int(res['opt_j'][k])) for k in range(len(pairs))])
            b.sync()
            res2 = b.results()
            check(res2, b.transcripts(res2))
            monkeypatch.delenv('PWLIB_STRIP_SPIN_LIMIT')
            res3 = b.run()                                  # default patience: the strips again
            check(res3, b.transcripts(res3))
            monkeypatch.setenv('PWLIB_STRIP_SPIN_LIMIT', '1')
            monkeypatch.setenv('PWLIB_NO_STRIP_REPAIR', '1')
            b.solve(); b.traceback(); b.sync()
            with pytest.raises(RuntimeError, match='abandoned'):
                b.results()
            monkeypatch.delenv('PWLIB_NO_STRIP_REPAIR')
            monkeypatch.delenv('PWLIB_STRIP_SPIN_LIMIT')
            res4 = b.run()
            check(res4, b.transcripts(res4))


BLASTISH = [[2, -3, -1, -3], [-3, 2, -3, -1], [-1, -3, 2, -3], [-3, -1, -3, 2]]       # transitions cost less than transversions
ASYM = [[5, -4, -2, 0], [-3, 4, -1, -6], [-2, -2, 6, -3], [1, -5, -4, 3]]             # asymmetric, one positive mismatch
WIDE = [[40, -30, -60, -35], [-30, 45, -25, -70], [-60, -25, 50, -20], [-35, -70, -20, 42]]   # range 120: unscaled form only


@pytest.mark.parametrize('mode,alntype,dr', [(1, 1, (-40, 35)), (1, 2, (-50, 50)), (1, 0, (-60, 60)), (0, 1, None),
                                             (0, 0, None), (0, 4, None), (0, 6, None)],
                         ids=['B_LOCAL', 'B_OVERLAP', 'B_GLOBAL', 'LOCAL', 'GLOBAL', 'OVERLAP', 'END_ANCHORED_OVERLAP'])
def test_integer_substitution_matrix_on_the_packed_kernels(oracle, mode, alntype, dr):
    """A 4 x 4 (3 x 3, 2 x 2) integer substitution matrix runs on the packed 16-bit kernels (rows of bytes + one byte permute
    per cell pair) instead of the generic kernel: the kernel name says so, and every record and transcript equals the forced
    generic kernel's and, on a sample, the oracle's (_alnchoice_M: subst_scores[o][m], 'M' iff the letters are equal)."""
    from biseqt_amd import _pwlib as W
    from biseqt_amd import synth
    rng = synth.rng_for(3400 + 10 * mode + alntype)
    for subst, L, go, ge in ((BLASTISH, 4, -5, -2), (ASYM, 4, 0, -3), (WIDE, 4, -20, -5), ([[3, -2, -4], [-1, 2, -3], [-4, -3, 5]], 3, -2, -1),
                             ([[1, -1], [-2, 2]], 2, -1, -1)):
        pairs = _pairs(rng, 320, 0 if L == 3 else 20, 400 if mode else 200, L=L)
        if subst is WIDE:                                   # scores up to 50 per letter: keep min(X, Y) * 50 below 8000
            pairs = [(o[:150], m[:150]) for o, m in pairs]
        kw = dict(alnmode=mode, alntype=alntype, alphabet_len=L, subst_scores=subst, go_score=go, ge_score=ge)
        if dr is not None:
            kw['diag_range'] = dr
        name, dtype, res, txs, rcs = _run(pairs, **kw)
        nameg, _, resg, txsg, _ = _run(pairs, flags=W.PW_FLAG_FORCE_GENERIC, **kw)
        assert 'k_fill16' in name and 'matrix' in name, (name, kw)
        assert 'k_fill16' not in nameg, nameg
        assert (res == resg).all() and txs == txsg, (name, kw)
        okw = dict(L=L, mode=mode, alntype=alntype, diag_range=dr, subst=[[float(v) for v in row] for row in subst], go=go, ge=ge)
        _check_vs_oracle(oracle, pairs, res, txs, rcs, okw, 13, (name, kw))
        if (mode, alntype) in ((1, 1), (0, 1)):
            # the scores-times-4 form: running scores below 2048 and 4 x the matrix's range within a byte's 127
            smax, smin = max(map(max, subst)), min(map(min, subst))
            longest = max(min(len(o), len(m)) for o, m in pairs)
            assert ('x4' in name) == (longest * max(smax, 0) <= 2047 and 4 * (smax - smin) <= 127), name


def test_matrices_the_packed_kernels_do_not_take(oracle):
    """Outside the admission (five letters, a range above 127, every score positive, go > 0) a matrix batch keeps the generic
    kernel -- and still equals the oracle."""
    from biseqt_amd import synth
    rng = synth.rng_for(3477)
    cases = [([[1 if i == j else -1 - (i + j) % 3 for j in range(5)] for i in range(5)], 5, -2),
             ([[100, -90, 0, 0], [-90, 100, 0, 0], [0, 0, 100, -90], [0, 0, -90, 100]], 4, -2),
             ([[5, 1, 2, 1], [1, 5, 1, 2], [2, 1, 5, 1], [1, 2, 1, 5]], 4, -2),
             (BLASTISH, 4, 2)]
    for subst, L, go in cases:
        pairs = _pairs(rng, 300, 10, 60, L=L)
        kw = dict(alnmode=1, alntype=1, alphabet_len=L, diag_range=(-20, 20), subst_scores=subst, go_score=go, ge_score=-1)
        name, dtype, res, txs, rcs = _run(pairs, **kw)
        assert 'k_fill16' not in name, name
        okw = dict(L=L, mode=1, alntype=1, diag_range=(-20, 20), subst=[[float(v) for v in row] for row in subst], go=go, ge=-1)
        _check_vs_oracle(oracle, pairs, res, txs, rcs, okw, 17, name)


@pytest.mark.parametrize('alntype,tag', [(2, ', 5>'), (3, ', 4>')], ids=['START_ANCHORED', 'END_ANCHORED'])
def test_anchored_types_on_their_packed_kernels(oracle, alntype, tag):
    """START_ANCHORED and END_ANCHORED (_alnchoice_B :161-209, _std_find_optimal :303-360) have their own packed
    instantiations (WaveFill16 rules 5 / 4): named, equal to the 32-bit kernels for every pair, and to the oracle on a sample
    -- many pairs (throughput layout), a few (latency layouts), with related, unrelated, suffix-prefix and empty pairs."""
    from biseqt_amd import _pwlib as W
    from biseqt_amd import synth
    rng = synth.rng_for(3500 + alntype)
    for n, (match, mismatch, go, ge) in ((1200, (2, -3, -4, -1)), (300, (1, -1, 0, -1)), (40, (5, -4, -10, -1)), (1100, (-1, 1, -2, -1))):
        pairs = _pairs(rng, n, 0, 500 if n > 100 else 1500)
        kw = dict(alnmode=0, alntype=alntype, alphabet_len=4, match_score=match, mismatch_score=mismatch, go_score=go, ge_score=ge)
        name, dtype, res, txs, rcs = _run(pairs, **kw)
        name32, _, res32, txs32, _ = _run(pairs, flags=W.PW_FLAG_NO_PACKED16, **kw)
        assert 'k_fill16' not in name32, name32
        if n > 256 and mismatch <= 0:
            assert 'k_fill16' in name and tag in name, (name, n)
        elif n > 256:
            # a mismatch score above 0 does not reach the plain packed form (off-table letters score it: DESIGN.md section 5),
            # and the anchored rules have no matrix form
            assert 'k_fill16' not in name, (name, n)
        assert (res == res32).all() and txs == txs32, (name, kw)
        okw = dict(L=4, mode=0, alntype=alntype, match=match, mismatch=mismatch, go=go, ge=ge)
        _check_vs_oracle(oracle, pairs, res, txs, rcs, okw, max(1, n // 40), (name, kw))


def test_float_matrices_on_the_fast_f64_kernels(oracle):
    """A full matrix of float scores (what a log-odds model with unequal substitution probabilities gives) used to take the
    generic f64 kernel; the fast f64 kernels now read every substitution score from the table in LDS, so it runs on them:
    named (not the generic instantiation `<double, BK, false, true, true>`), equal to the forced generic kernel for every pair
    and to the oracle on samples -- alphabets of 4 and 20 letters; 40 letters (table too large for the LDS copy) stay generic."""
    from biseqt_amd import _pwlib as W
    from biseqt_amd import synth
    rng = synth.rng_for(3700)
    for L, fast in ((4, True), (20, True), (40, False)):
        P = rng.uniform(0.02, 1.0, size=(L, L)) + np.eye(L) * 4
        P = P / P.sum(axis=1, keepdims=True)
        subst = [[float(np.log(0.9) + np.log(P[i][j]) - np.log(1.0 / L)) for j in range(L)] for i in range(L)]
        go, ge = float(np.log(0.05) - np.log(0.1)), float(np.log(0.1))
        for mode, alntype, dr in ((1, 1, (-30, 30)), (0, 0, None), (1, 2, (-40, 40))):
            pairs = _pairs(rng, 300, 10, 220, L=L)
            kw = dict(alnmode=mode, alntype=alntype, alphabet_len=L, subst_scores=subst, go_score=go, ge_score=ge)
            if dr is not None:
                kw['diag_range'] = dr
            name, dtype, res, txs, rcs = _run(pairs, **kw)
            nameg, _, resg, txsg, _ = _run(pairs, flags=W.PW_FLAG_FORCE_GENERIC, **kw)
            assert dtype == 'f64' and 'double' in name, name
            assert name.endswith('false, true, true>') != fast or 'k_fill_mw' in name or 'k_fill_tile' in name, (name, L)
            assert (res == resg).all() and txs == txsg, (name, L, mode)
            okw = dict(L=L, mode=mode, alntype=alntype, diag_range=dr, subst=subst, go=go, ge=ge)
            _check_vs_oracle(oracle, pairs, res, txs, rcs, okw, 17, (name, L))


def test_protein_style_integer_matrix_on_the_fast_kernels(oracle):
    """An integer matrix over 20 letters (BLOSUM-style: the packed kernels' byte rows hold 4 letters only) runs on the fast
    32-bit kernels, which read every substitution score from a table in LDS, not on the generic kernel: equal to the forced
    generic kernel for every pair and to the oracle on samples; also through the workgroup (wide band) and tiled kernels."""
    from biseqt_amd import _pwlib as W
    from biseqt_amd import synth
    rng = synth.rng_for(3800)
    L = 20
    S = rng.integers(-4, 4, size=(L, L))
    S = (S + S.T) // 2
    S[np.arange(L), np.arange(L)] = rng.integers(4, 12, size=L)
    subst = [[int(v) for v in row] for row in S]
    for mode, alntype, dr, n, lo, hi, flags in ((1, 1, (-30, 30), 300, 10, 300, 0), (0, 0, None, 300, 10, 200, 0), (1, 2, (-40, 40), 300, 50, 400, 0),
                                                (0, 1, None, 3, 1200, 1500, 0), (0, 1, None, 2, 500, 700, W.PW_FLAG_FORCE_TILED)):
        pairs = _pairs(rng, n, lo, hi, L=L)
        kw = dict(alnmode=mode, alntype=alntype, alphabet_len=L, subst_scores=subst, go_score=-10, ge_score=-1)
        if dr is not None:
            kw['diag_range'] = dr
        name, dtype, res, txs, rcs = _run(pairs, flags=flags, **kw)
        assert dtype == 'i32' and not name.endswith('false, true, true>'), name
        if not flags:
            nameg, _, resg, txsg, _ = _run(pairs, flags=W.PW_FLAG_FORCE_GENERIC, **kw)
            assert (res == resg).all() and txs == txsg, (name, nameg)
        okw = dict(L=L, mode=mode, alntype=alntype, diag_range=dr, subst=[[float(v) for v in row] for row in subst], go=-10, ge=-1)
        _check_vs_oracle(oracle, pairs, res, txs, rcs, okw, max(1, n // 20), name)


def test_positive_mismatch_scores_and_long_waits(oracle):
    """Regression (round 3, found by tests/micro/fuzz_gpu.py, seed 9551): the API accepts a mismatch score above 0.  The plain
    form of the packed kernels scores letters outside a sequence as a mismatch, so a diagonal that waits long for its first
    cell crept up from the 16-bit sentinel and started from a positive phantom score -- a wrong end cell, and the walk from it
    left the mask plane (a GPU memory fault).  Such scores now take the packed matrix form where it applies (off-table letters
    score the matrix minimum, <= 0) and the 32-bit kernels otherwise; and the walker refuses cells beyond (X, Y).  Cases: the
    pair the fuzz saved (3673-diagonal band, 20 letters, 1 / 6 / -5 / -2), the same shape over 4 letters with -1 / 2 (matrix
    form), and batches of ordinary shapes with both score sets."""
    import os
    d = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'regress', 'mw_mismatch_9551.npz'))
    o, m = d['o'], d['m']
    band = tuple(int(v) for v in d['band'])
    kw = dict(alnmode=1, alntype=1, alphabet_len=20, match_score=1, mismatch_score=6, go_score=-5, ge_score=-2)
    okw = dict(L=20, mode=1, alntype=1, match=1, mismatch=6, go=-5, ge=-2, diag_range=band)
    for n in (1, 3):
        name, _, res, txs, rcs = _run([(o, m)] * n, diag_range=[band] * n, **kw)
        assert 'k_fill16' not in name, name
        _check_vs_oracle(oracle, [(o, m)] * n, res, txs, rcs, okw, 1, ('saved pair', n))
    rng = np.random.default_rng(20261008)
    o4 = rng.integers(0, 4, 524).astype(np.uint8)
    m4 = rng.integers(0, 4, 3656).astype(np.uint8)
    for sc, expect_matrix in (((-1, 2, -2, -1), True), ((1, 6, -5, -2), False), ((2, 3, -1, -1), False)):
        kw4 = dict(alnmode=1, alntype=1, alphabet_len=4, match_score=sc[0], mismatch_score=sc[1], go_score=sc[2], ge_score=sc[3])
        okw4 = dict(L=4, mode=1, alntype=1, match=sc[0], mismatch=sc[1], go=sc[2], ge=sc[3], diag_range=band)
        name, _, res, txs, rcs = _run([(o4, m4)], diag_range=[band], **kw4)
        assert ('k_fill16' in name) == expect_matrix and (not expect_matrix or 'matrix' in name), (sc, name)
        _check_vs_oracle(oracle, [(o4, m4)], res, txs, rcs, okw4, 1, ('4 letters', sc))
        for mode, alntype, dr in ((1, 1, (-300, 280)), (1, 2, (-150, 200)), (0, 1, None), (0, 0, None)):
            pairs = _pairs(rng, 40, 150, 600)
            kwb = dict(kw4, alnmode=mode, alntype=alntype)
            okwb = dict(okw4, mode=mode, alntype=alntype)
            okwb.pop('diag_range')
            if dr is not None:
                kwb['diag_range'] = dr
                okwb['diag_range'] = dr
            name, _, res, txs, rcs = _run(pairs, **kwb)
            assert 'k_fill16' not in name or 'matrix' in name, (sc, mode, alntype, name)
            _check_vs_oracle(oracle, pairs, res, txs, rcs, okwb, 1, (sc, mode, alntype))
