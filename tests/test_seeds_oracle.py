"""The seeds oracle against the known answers the reference's own tests and docstrings hold
(reference tests/test_seeds.py, tests/test_kmers.py, biseqt/seeds.py:10-16), restated."""
from itertools import product

import numpy as np
import pytest

from oracle import seeds_oracle as SO

ACGT = {c: i for i, c in enumerate('ACGT')}


def parse(s):
    return [ACGT[c] for c in s]


def test_docstring_example():                       # seeds.py:10-16
    rows, sc = SO.seed_rows(parse('TAAGCGT'), parse('GGCGTAA'), 3, 4)
    assert not sc and SO.seeds(rows, sc) == [(4, 2), (3, 1), (0, 4)]


def test_coordinate_change():                       # tests/test_seeds.py:10-31
    pairs = {(0, 0): (0, 0), (0, 1): (-1, 1), (1, 0): (1, 1), (1, 1): (0, 2)}
    for (i, j), (d, a) in pairs.items():
        assert SO.to_diagonal_coordinates(i, j) == (d, a) and SO.to_ij_coordinates(d, a) == (i, j)
    assert SO.to_ij_coordinates_seg(((-2, 2), (0, 2))) == ((0, 2), (0, 2))


@pytest.mark.parametrize('wordlen', [5, 15])
def test_index_seeds(wordlen):                      # tests/test_seeds.py:92-120
    S = parse('G' * wordlen); T = parse('TC') + S
    rows, sc = SO.seed_rows(S, T, wordlen, 4)
    assert SO.seeds(rows, sc) == [(0, 2)]
    rows, sc = SO.seed_rows(T, S, wordlen, 4)
    assert SO.seeds(rows, sc) == [(2, 0)]
    S = parse('A' * 5 * wordlen); T = parse('A' * 10 * wordlen)
    rows, sc = SO.seed_rows(S, T, wordlen, 4)
    assert len(SO.seeds(rows, sc)) == (len(S) - wordlen + 1) * (len(T) - wordlen + 1)
    rows, sc = SO.seed_rows(S, S, wordlen, 4)
    assert sc and len(SO.seeds(rows, sc)) == (len(S) - wordlen + 1) ** 2


@pytest.mark.parametrize('wordlen', [5, 15])
def test_index_integrity(wordlen):                  # tests/test_seeds.py:60-89
    S = parse('AAACCCGGGCAAGCC')
    T = parse('T' * 2 * wordlen) + S + parse('T' * 2 * wordlen)
    rows, sc = SO.seed_rows(S, T, wordlen, 4)
    assert len(SO.seeds(rows, sc)) == len(S) - wordlen + 1


@pytest.mark.parametrize('wordlen', [5, 15])
def test_seed_counts(wordlen):                      # tests/test_seeds.py:123-171
    rng = np.random.default_rng(wordlen)
    S = rng.integers(0, 4, 5 * wordlen).tolist(); T = rng.integers(0, 4, 5 * wordlen).tolist()
    rows, sc = SO.seed_rows(S, T, wordlen, 4)
    assert len(SO.seeds(rows, sc)) == SO.seed_count(rows)
    for d in range(-wordlen, wordlen):
        band = (d - wordlen, d + wordlen)
        assert len(SO.seeds(rows, sc, d_band=band)) == SO.seed_count(rows, d_band=band)
    rows, sc = SO.seed_rows(S, list(S), wordlen, 4)
    assert sc and SO.seed_count(rows, d_band=(0, 0)) == len(S) - wordlen + 1
    assert len(SO.seeds(rows, sc, d_band=(0, 0), exclude_trivial=True)) == 0
    S = parse('T' * wordlen + 'G' * wordlen); T = parse('G' * wordlen + 'T' * wordlen)
    rows, sc = SO.seed_rows(S, T, wordlen, 4)
    assert SO.seed_count(rows) == 2
    assert SO.seed_count(rows, d_band=(-wordlen - 1, -wordlen + 1)) == 1
    assert SO.seed_count(rows, d_band=(wordlen - 1, wordlen + 1)) == 1
    assert SO.seed_count(rows, a_band=(wordlen, wordlen)) == 2
    rows, sc = SO.seed_rows(S, parse('C' * wordlen + 'A' * wordlen), wordlen, 4)
    assert SO.seed_count(rows) == 0


@pytest.mark.parametrize('L', [4, 3])
@pytest.mark.parametrize('wordlen', [3, 6, 9])
def test_kmer_as_int(L, wordlen):                   # tests/test_kmers.py:22-31, kmers.py:166-168 (AGA -> 8)
    ints = [SO.kmer_as_int(k, L) for k in product(range(L), repeat=wordlen)]
    assert len(set(ints)) == L ** wordlen
    assert SO.kmer_as_int(parse('AGA'), 4) == 8


@pytest.mark.parametrize('L', [4, 3])
@pytest.mark.parametrize('wordlen', [3, 6, 9])
def test_kmer_masks(L, wordlen):                    # tests/test_kmers.py:34-53
    assert all(k is None for k in SO.as_kmer_seq([0] * 10, wordlen, L, mask=[{0}]))
    ks = SO.as_kmer_seq([0] * 10 + [1], wordlen, L, mask=[{0}])
    assert sum(k for k in ks if k is not None) == 1
    rng = np.random.default_rng(5)
    S = rng.integers(1, 3, 10).tolist() + [0]
    ks = SO.as_kmer_seq(S, wordlen, L, mask=[{1}, {2}, {1, 2}])
    assert sum(int(k is not None) for k in ks) == 1


def test_in_memory_variant_is_the_same_set():       # blot.py:607-620 against seeds.py:164-197
    rng = np.random.default_rng(9)
    S = rng.integers(0, 4, 300).tolist(); T = rng.integers(0, 4, 250).tolist()
    rows, sc = SO.seed_rows(S, T, 4, 4)
    assert sorted(SO.seeds(rows, sc)) == sorted(SO.seeds_by_mutant(S, T, 4, 4))
    rows, sc = SO.seed_rows(S, S, 4, 4)
    assert sorted(SO.seeds(rows, sc, exclude_trivial=True)) == sorted(SO.seeds_by_mutant(S, S, 4, 4))


def _rows_ok(rec, key, rows):
    import hashlib
    rows = [(int(i), int(j)) for i, j in rows]
    assert len(rows) == rec[key + '_n'], (key, len(rows), rec[key + '_n'])
    assert hashlib.sha256(','.join('%d:%d' % r for r in rows).encode()).hexdigest() == rec[key + '_sha256'], key
    if key in rec:
        assert [list(r) for r in rows] == rec[key], key


def test_oracle_equals_the_reference_seed_lists():
    """Seed lists and band counts computed by the reference's OWN in-memory enumeration (`WordBlotOverlapRef.seeds` /
    `seed_count`, blot.py:607-642; fixture generated by tests/golden/make_logodds_golden.py importing the reference) for
    64 seeded pairs -- related, unrelated, repeats, two-letter sequences, self comparisons, sequences shorter than the
    word: the oracle's enumeration must give the same rows in the same order, and its table rows the same set."""
    from tests.helpers import dec, load_golden
    recs = load_golden('seed_lists_reference.json.gz')
    assert len(recs) >= 50
    n_rows = 0
    for k, rec in enumerate(recs):
        S, T, w = dec(rec['S']), dec(rec['T']), rec['wordlen']
        got = SO.seeds_by_mutant(S, T, w, 4)
        _rows_ok(rec, 'seeds_ij', got)
        n_rows += len(got)
        if 'seeds_ij_with_trivial_n' in rec:
            _rows_ok(rec, 'seeds_ij_with_trivial', SO.seeds_by_mutant(S, T, w, 4, exclude_trivial=False))
        # the SQL-table presentation (seeds.py:117-197) lists the same seeds in another order
        rows, sc = SO.seed_rows(S, T, w, 4)
        assert sorted(SO.seeds(rows, sc, exclude_trivial=True)) == sorted(got), k
        for b in rec['band_counts']:
            (d0, d1), (a0, a1) = b['d_band'], b['a_band']
            da = [SO.to_diagonal_coordinates(i, j) for i, j in got]
            assert sum(1 for d, a in da if d0 <= d <= d1) == b['count_d'], k
            assert sum(1 for d, a in da if d0 <= d <= d1 and a0 <= a <= a1) == b['count_da'], k
    assert n_rows > 50000
