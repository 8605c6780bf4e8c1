"""Host-side logic mirroring the reference's own pure-Python tests: Alignment (tests/test_pw.py:10-31,
143-154, 188-261), Aligner argument validation (:75-78), Alphabet/Sequence, the log-odds formula,
the planner's cell counts and the workload generator.  No GPU."""
import math

import numpy as np
import pytest

from biseqt_amd import synth
from biseqt_amd.pw import Aligner, Alignment, BANDED_MODE
from biseqt_amd.sequence import Alphabet, Sequence
from biseqt_amd.stochastics import MutationProcess, rand_seq


def test_projected_aln_len():
    assert Alignment.projected_len('MMM', on='origin') == 3
    assert Alignment.projected_len('MMM', on='mutant') == 3
    assert Alignment.projected_len('SMS', on='origin') == 3
    assert Alignment.projected_len('SMS', on='mutant') == 3
    assert Alignment.projected_len('DMS', on='origin') == 3
    assert Alignment.projected_len('DMS', on='mutant') == 2
    assert Alignment.projected_len('IMS', on='origin') == 2
    assert Alignment.projected_len('IMS', on='mutant') == 3


@pytest.mark.parametrize('alphabet', [Alphabet('ACGT'), Alphabet(['00', '01'])],
                         ids=['one letter alphabet', 'two letter alphabet'])
def test_alignment_constructor_assertions(alphabet):
    S = alphabet.parse(alphabet[0] * 10)
    with pytest.raises(AssertionError):
        Alignment(S, S, 'MSSST')                      # illegal character
    with pytest.raises(AssertionError):
        Alignment(S, S, 'M', origin_start=len(S))     # illegal starting point
    with pytest.raises(AssertionError):
        Alignment(S, S, 'MM', origin_start=len(S) - 1)  # transcript too long
    # banded diag_range outside the table is refused in Python, before any C call (test_pw.py:75-78)
    with pytest.raises(AssertionError):
        Aligner(S, S, alnmode=BANDED_MODE, diag_range=(-len(S) - 1, 0))
    with pytest.raises(AssertionError):
        Aligner(S, S, alnmode=BANDED_MODE, diag_range=(0, len(S) + 1))
    with pytest.raises(AssertionError):
        Aligner(S, S, alnmode=BANDED_MODE, alntype=5)


def test_alignment_eq_ignores_score_and_calculate_score():
    A = Alphabet('ACGT')
    S, T = A.parse('AAACGCGT'), A.parse('AACGCCTT')
    a = Alignment(S, T, 'MMDMMMIDMI', score=6.)
    b = Alignment(S, T, 'MMDMMMIDMI', score=None)
    assert a == b
    subst = [[1 if i == j else 0 for j in range(4)] for i in range(4)]
    assert a.calculate_score(subst, 0, 0) == 6.
    assert a.calculate_score(subst, -5, -2) == 6. + 4 * (-5) + 4 * (-2)      # four gap runs of length 1


def test_pw_truncate_to_matches():
    A = Alphabet('ACGT')
    S = A.parse('A' * 10 + 'T' * 10 + 'A' * 10)
    T = A.parse('T' * 30)
    aln = Alignment(S, T, 'S' * 10 + 'M' * 10 + 'S' * 10)
    t = aln.truncate_to_match()
    assert t.transcript == 'M' * 10 and t.origin_start == 10 and t.mutant_start == 10


def test_pw_render_basic():
    A = Alphabet('ACGT')
    S = A.parse('AACT')
    aln = Alignment(S, S, 'M' * len(S))
    assert aln.render_term(colored=False).count('\033') == 0
    assert aln.render_term(colored=True).count('\033') > 0
    with pytest.raises(AssertionError):
        aln.render_term(margin=-1)
    with pytest.raises(AssertionError):
        aln.render_term(term_width=5)
    aln = Alignment(S + S, S + S, 'M' * len(S), origin_start=len(S))
    assert '[%d]' % len(S) in aln.render_term(margin=0, colored=False)
    assert '[%d]' % (len(S) - 1) in aln.render_term(margin=1, colored=False)
    full_margin = aln.render_term(margin=30, colored=False)
    assert str(S) + '.' * len(S) in full_margin
    assert len(set(len(l) for l in full_margin.rstrip().split('\n'))) == 1
    aln = Alignment(S + S, A.parse('AGT'), 'MSDM', origin_start=len(S))
    with_del = aln.render_term(colored=False)
    assert 'AG-T' in with_del
    lines = with_del.rstrip().split('\n')
    assert lines[0].index('C') == lines[1].index('-')
    aln.render_term(colored=True)
    aln = Alignment(S + S, A.parse('AACGT'), 'MMMIM', origin_start=len(S))
    with_ins = aln.render_term(colored=False)
    assert 'AAC-T' in with_ins
    lines = with_ins.rstrip().split('\n')
    assert lines[0].index('-') == lines[1].index('G')
    assert '-----' in str(Alignment(A.parse('A' * 10), A.parse('A' * 5), 'MMMMMDDDDD'))


def test_pw_render_width_and_long_letters():
    A = Alphabet('ACGT')
    N = 100
    S = A.parse('A' * (2 * N))
    term_width = N // 2
    aln = Alignment(S, S, 'M' * N, origin_start=N)
    render = aln.render_term(margin=2 * N, colored=False, term_width=term_width)
    line_lens = [len(l) for l in render.rstrip().split('\n')]
    assert all(length <= term_width for length in line_lens)
    assert any(length == term_width for length in line_lens)
    assert len(set(line_lens)) <= 2
    B = Alphabet(['00', '11'])
    assert '--11' in Alignment(B.parse('0011'), B.parse('11'), 'DM').render_term(colored=False)


def test_alphabet_and_sequence():
    A = Alphabet('ACGT')
    S = A.parse('AACTTCG')
    assert S.contents == (0, 0, 1, 3, 3, 1, 2)
    assert str(S[:3]) == 'AAC' and S[2] == 1 and len(S) == 7
    assert len(S.content_id) == 40
    assert str(A.transform(A.parse('AGGGT'), mappings=['AT', 'CG'])) == 'TCCCA'
    assert str(S.reverse()) == 'GCTTCAA'
    assert S + 'AC' == A.parse('AACTTCGAC') and S + S[:1] == A.parse('AACTTCGA')
    B = Alphabet(['A1', 'A2', 'A3', 'A4'])
    assert len(B.parse('A1A1A3A2')) == 4
    with pytest.raises(AssertionError):
        Alphabet(['A', 'BC'])
    with pytest.raises(AssertionError):
        Sequence(A, (0, 4))
    assert bool(Sequence(A)) is False


def test_log_odds_scores_formula_and_signs():
    A = Alphabet('ACGT')
    M = MutationProcess(A, subst_probs=.2, go_prob=.1, ge_prob=.3)
    S, (go, ge) = M.log_odds_scores()
    assert S[0][0] == math.log(.7) + math.log(.8) - math.log(.25)
    assert S[0][1] == math.log(.7) + math.log(.2 / 3) - math.log(.25)
    assert go == math.log(.1) - math.log(.3) and ge == math.log(.3)
    assert S[0][0] > 0 > S[0][1] and go < 0 and ge < 0
    M2 = MutationProcess(A, subst_probs=.1, go_prob=.2, ge_prob=.2)
    assert M2.log_odds_scores()[1][0] == 0          # linear model: gap-open score 0
    with pytest.raises(AssertionError):
        MutationProcess(A, subst_probs=.1, go_prob=.3, ge_prob=.2)


def test_log_odds_scores_equal_the_reference_fixture():
    """`MutationProcess.log_odds_scores` against values computed by the reference's own function
    (/root/reference/biseqt/stochastics.py:234-310; tests/golden/logodds_reference.json, generated by
    tests/golden/make_logodds_golden.py importing it): the five noise levels of the reference's tests/test_pw.py:106,
    alphabets of 2 / 4 / 20 letters, linear and affine gap models, uniform and own null hypotheses -- every score equal
    as a bit pattern.  These are the float scores the f64 DP path is fed with."""
    import json
    import os
    with open(os.path.join(os.path.dirname(__file__), 'golden', 'logodds_reference.json')) as f:
        recs = json.load(f)['records']
    assert len(recs) >= 60
    for k, rec in enumerate(recs):
        A = Alphabet(rec['letters'])
        sp = rec['subst_probs']
        sp = float.fromhex(sp) if isinstance(sp, str) else [[float.fromhex(v) for v in row] for row in sp]
        M = MutationProcess(A, subst_probs=sp, go_prob=float.fromhex(rec['go_prob']), ge_prob=float.fromhex(rec['ge_prob']))
        null = None if rec['null'] is None else [float.fromhex(v) for v in rec['null']]
        S, (go, ge) = M.log_odds_scores(null_hypothesis=null) if null is not None else M.log_odds_scores()
        assert [[float(v).hex() for v in row] for row in S] == rec['subst_scores'], (k, rec['tag'])
        assert (float(go).hex(), float(ge).hex()) == (rec['go_score'], rec['ge_score']), (k, rec['tag'])


def test_mutate_transcript_is_consistent():
    A = Alphabet('ACGT')
    rng = np.random.default_rng(3)
    M = MutationProcess(A, subst_probs=.1, go_prob=.1, ge_prob=.3, rng=rng)
    S = rand_seq(A, 200, rng=rng)
    T, tx = M.mutate(S)
    aln = Alignment(S, T, tx)                         # constructor checks the projected lengths
    assert Alignment.projected_len(tx, on='origin') == len(S)
    assert Alignment.projected_len(tx, on='mutant') == len(T)
    assert all((S[i] == T[j]) == (op == 'M') for op, i, j in _walk(tx) if op in 'MS')
    assert aln.transcript == tx


def _walk(tx):
    i = j = 0
    for op in tx:
        yield op, i, j
        i += op in 'MSD'
        j += op in 'MSI'


def test_cell_counts_match_oracle(oracle):
    rng = np.random.default_rng(0)
    for _ in range(200):
        X, Y = int(rng.integers(0, 60)), int(rng.integers(0, 60))
        lo, hi = sorted(rng.integers(-Y - 3, X + 4, 2).tolist())
        want = oracle.cells(X, Y, mode=1, alntype=oracle.B_LOCAL, diag_range=(lo, hi))
        got = synth.banded_cells(X, Y, lo, hi)
        assert got == max(want, 0) or (want < 0 and got == 0), (X, Y, lo, hi, got, want)
    assert synth.banded_cells(2000, 2000, -200, 200) == 762201
    assert synth.std_cells(1000, 1000) == 1002001


def test_synthetic_batch_is_deterministic():
    o1, m1 = synth.pair_batch(2, 5, 300)
    o2, m2 = synth.pair_batch(2, 5, 300)
    assert all((a == b).all() for a, b in zip(o1, o2)) and all((a == b).all() for a, b in zip(m1, m2))
    assert all(abs(len(m) - 300) < 60 for m in m1)


def test_rand_read_and_noisy_read():
    """reference tests/test_stochastics.py:19-62 (lossless reads, coverage) and :203-232 (noisy reads)."""
    from biseqt_amd.stochastics import rand_read
    rng = np.random.default_rng(4)
    A = Alphabet('ACGT')
    S = rand_seq(A, 100, rng=rng)
    assert len(list(rand_read(S, len_mean=len(S) / 2, expected_coverage=10, rng=rng))) == 20
    with pytest.raises(AssertionError):
        next(rand_read(S, len_mean=200, num=1))
    with pytest.raises(AssertionError):
        next(rand_read(S, len_mean=50, num=1, expected_coverage=1))
    assert sum(1 for _ in rand_read(S, len_mean=50, num=10, rng=rng)) == 10
    assert sum(1 for _ in rand_read(S, len_mean=50, rng=rng)) == 1
    read, pos = next(rand_read(S, len_mean=40, num=1, rng=rng))
    assert S[pos:pos + len(read)] == read
    S3 = A.parse('ACT' * 100)
    reads = list(rand_read(S3, len_mean=100, len_sd=10, num=100, rng=rng))
    assert len(set(len(r) for r, _ in reads)) > 1
    assert 50 < sum(len(r) for r, _ in reads) / 100. < 150
    A2 = Alphabet(['00', '01'])
    S2 = A2.parse('01' * 10)
    r, _ = next(rand_read(S2, len_mean=1, len_sd=1e-9, num=1, rng=rng))
    assert r == A2.parse('01')
    M = MutationProcess(A, subst_probs=.1, go_prob=.1, ge_prob=.2, rng=rng)
    out = list(M.noisy_read(S, len_mean=30, num=5))
    assert len(out) == 5
    for noisy, start, tx in out:
        n_orig = sum(1 for c in tx if c in 'MSD')
        assert S[start:start + n_orig].contents == S.contents[start:start + n_orig]
        assert len(noisy) == sum(1 for c in tx if c in 'MSI')
