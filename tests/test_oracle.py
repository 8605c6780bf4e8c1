"""The oracle (oracle/pw_oracle.c) against the golden vectors generated from the compiled reference,
and -- where the compiled reference itself is present (this container; oracle/_ref travels to the GPU
box as a prebuilt file) -- against the reference live on random problems."""
import os

import numpy as np
import pytest

from tests.helpers import check_against_expect, dec, kw_of, load_golden


@pytest.mark.parametrize('name', ['known_answers.json', 'random_matrix.json.gz', 'float_logodds.json',
                                  'config_sized.json'])
def test_oracle_matches_golden(oracle, name):
    recs = load_golden(name)
    assert len(recs) > 0
    for k, rec in enumerate(recs):
        got = oracle.solve(dec(rec['origin']), dec(rec['mutant']), **kw_of(rec))
        if got.get('maskrule_ok') is not None:
            assert got['maskrule_ok'], (name, k, 'mask-only traceback rule')
        check_against_expect(got, rec['expect'], where='%s[%d]' % (name, k))


def test_reference_known_answers(oracle):
    """The assertions of the reference's tests/test_pw.py:33-103, restated on the oracle."""
    S, junk = [0] * 10, [1] * 10
    assert oracle.solve(S, S, L=4)['transcript'] == 'M' * 10                       # :33-36
    r = oracle.solve(S, S, L=4, want_table=True)
    H = r['H'].reshape(11, 11)
    assert H[1:, 1:].max() == H[-1, -1]                                            # :41-42
    assert oracle.solve(S, S[:5], L=4)['transcript'].count('D') == 5               # :44-48
    r = oracle.solve(S + junk, junk + S, L=4, alntype=oracle.LOCAL)                # :52-58
    assert (r['transcript'], r['origin_idx'], r['mutant_idx']) == ('M' * 10, 0, 10)
    assert oracle.solve(S, junk, L=4, alntype=oracle.LOCAL)['opt'] == (-1, -1)     # :60-62
    assert oracle.solve(S + junk, junk + S, L=4, alntype=oracle.OVERLAP)['transcript'] == 'M' * 10
    r = oracle.solve(S, S, L=4, mode=1, alntype=oracle.B_GLOBAL, diag_range=(0, 0))  # :80-83
    assert r['transcript'] == 'M' * 10
    r = oracle.solve(S + junk, junk + S, L=4, mode=1, alntype=oracle.B_OVERLAP,
                     diag_range=(-20, 20), ge=-1)                                  # :85-92
    assert (r['transcript'], r['origin_idx'], r['mutant_idx']) == ('M' * 10, 0, 10)


def test_docstring_example_and_non_gotoh(oracle):
    A = 'ACGT'
    e = lambda s: [A.index(c) for c in s]   # noqa: E731
    r = oracle.solve(e('AAACGCGT'), e('AACGCCTT'), L=4)                            # pw.py:11-21
    assert (r['transcript'], r['score']) == ('MMDMMMIDMI', 6.0)
    r = oracle.solve([2, 0, 2, 0, 3, 3, 0], [0, 0, 0], L=4, match=2, mismatch=-3, go=-4, ge=-1)
    assert r['score'] == -11.0     # textbook affine (Gotoh) would give -7: SURVEY.md section 7


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(__file__), '..', 'oracle', '_ref',
                                                    'pwlib_ref.so')),
                    reason='compiled reference (oracle/_ref) not present')
def test_oracle_matches_compiled_reference_live(capfd):
    from oracle import check_vs_ref, ref_driver
    rng = np.random.default_rng(2024)
    reflib = ref_driver.load()
    for t in range(600):
        origin, mutant, kw = check_vs_ref.random_problem(rng)
        errs, _ = check_vs_ref.compare(origin, mutant, kw, reflib)
        assert not errs, (origin, mutant, kw, errs[:3])
    import ctypes
    ctypes.CDLL(None).fflush(None)
    capfd.readouterr()     # drop the reference's stdout chatter (band clamp messages)
