"""GPU seeds (include/pw_seeds.h through biseqt_amd.seeds / kmers) against the seeds oracle and against the known
answers of the reference's tests (tests/test_seeds.py, tests/test_kmers.py, seeds.py:10-16)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def A():
    from biseqt_amd.sequence import Alphabet
    return Alphabet('ACGT')


def _seq(A, contents):
    from biseqt_amd.sequence import Sequence
    return Sequence(A, tuple(int(c) for c in contents))


def test_docstring_example(A):
    from biseqt_amd.seeds import SeedIndex
    S, T = A.parse('TAAGCGT'), A.parse('GGCGTAA')
    assert list(SeedIndex(S, T, wordlen=3, alphabet=A).seeds()) == [(4, 2), (3, 1), (0, 4)]


def test_coordinate_change():
    from biseqt_amd.seeds import SeedIndex
    pairs = {(0, 0): (0, 0), (0, 1): (-1, 1), (1, 0): (1, 1), (1, 1): (0, 2)}
    for (i, j), (d, a) in pairs.items():
        assert SeedIndex.to_diagonal_coordinates(i, j) == (d, a)
        assert SeedIndex.to_ij_coordinates(d, a) == (i, j)
    (i_start, i_end), (j_start, j_end) = SeedIndex.to_ij_coordinates_seg(((-2, 2), (0, 2)))
    assert i_start == j_start == 0 and i_end == j_end == 2


@pytest.mark.parametrize('wordlen', [5, 15])
def test_index_seeds_and_integrity(A, wordlen):      # reference tests/test_seeds.py:60-120
    from biseqt_amd.seeds import SeedIndex
    kw = dict(alphabet=A, wordlen=wordlen)
    S = A.parse('G' * wordlen)
    T = A.parse('TC' + 'G' * wordlen)
    assert list(SeedIndex(S, T, **kw).seeds()) == [(0, 2)] and list(SeedIndex(T, S, **kw).seeds()) == [(2, 0)]
    S = A.parse('A' * 5 * wordlen)
    T = A.parse('A' * 10 * wordlen)
    assert len(list(SeedIndex(S, T, **kw).seeds())) == (len(S) - wordlen + 1) * (len(T) - wordlen + 1)
    assert len(list(SeedIndex(S, S, **kw).seeds())) == (len(S) - wordlen + 1) ** 2
    S = A.parse('AAACCCGGGCAAGCC')
    T = A.parse('T' * 2 * wordlen + 'AAACCCGGGCAAGCC' + 'T' * 2 * wordlen)
    assert len(list(SeedIndex(S, T, **kw).seeds())) == len(S) - wordlen + 1


@pytest.mark.parametrize('wordlen', [5, 15])
def test_seed_counts(A, wordlen):                    # reference tests/test_seeds.py:123-171
    from biseqt_amd.seeds import SeedIndex
    kw = dict(alphabet=A, wordlen=wordlen)
    rng = np.random.default_rng(wordlen)
    S = _seq(A, rng.integers(0, 4, 5 * wordlen))
    T = _seq(A, rng.integers(0, 4, 5 * wordlen))
    idx = SeedIndex(S, T, **kw)
    assert len(list(idx.seeds())) == idx.seed_count()
    for d in range(-wordlen, wordlen):
        band = (d - wordlen, d + wordlen)
        assert len(list(idx.seeds(d_band=band))) == idx.seed_count(d_band=band)
    idx = SeedIndex(S, _seq(A, S.contents), **kw)
    assert idx.seed_count(d_band=(0, 0)) == len(S) - wordlen + 1
    assert len(list(idx.seeds(d_band=(0, 0), exclude_trivial=True))) == 0
    S = A.parse('T' * wordlen + 'G' * wordlen)
    T = A.parse('G' * wordlen + 'T' * wordlen)
    idx = SeedIndex(S, T, **kw)
    assert idx.seed_count() == 2
    assert idx.seed_count(d_band=(-wordlen - 1, -wordlen + 1)) == 1
    assert idx.seed_count(d_band=(wordlen - 1, wordlen + 1)) == 1
    assert idx.seed_count(a_band=(wordlen, wordlen)) == 2
    assert SeedIndex(S, A.parse('C' * wordlen + 'A' * wordlen), **kw).seed_count() == 0


def test_kmers_and_masks():                          # reference tests/test_kmers.py:22-68
    from biseqt_amd.kmers import as_kmer_seq, kmer_as_int
    from biseqt_amd.sequence import Alphabet, Sequence
    from oracle import seeds_oracle as SO
    for alphabet in (Alphabet('ACGT'), Alphabet(['00', '01', '11'])):
        L = len(alphabet)
        for wordlen in (3, 6, 9, 13, 23):
            if L ** wordlen >= 2 ** 62:
                continue
            rng = np.random.default_rng(wordlen)
            S = Sequence(alphabet, tuple(int(c) for c in rng.integers(0, L, 50)))
            ks = as_kmer_seq(S, wordlen)
            assert ks == SO.as_kmer_seq(list(S.contents), wordlen, L) and len(ks) == len(S) - wordlen + 1
            assert all(k == kmer_as_int(S.contents[p:p + wordlen], alphabet) for p, k in enumerate(ks))
        for wordlen in (3, 6, 9):
            S = Sequence(alphabet, tuple([0] * 10))
            assert all(k is None for k in as_kmer_seq(S, wordlen, mask=[set([0])]))
            S = Sequence(alphabet, tuple([0] * 10 + [1]))
            assert sum(k for k in as_kmer_seq(S, wordlen, mask=[set([0])]) if k is not None) == 1
            mask = [set([1]), set([2]), set([1, 2])]
            rng = np.random.default_rng(3)
            S = Sequence(alphabet, tuple([int(c) for c in rng.integers(1, 3, 10)] + [0]))
            ks = as_kmer_seq(S, wordlen, mask=mask)
            assert sum(int(k is not None) for k in ks) == 1
            assert ks == SO.as_kmer_seq(list(S.contents), wordlen, L, mask)


def test_random_pairs_vs_oracle(A):
    """Row order, rows, seeds(), band counts: identical to the oracle on random and adversarial pairs."""
    from biseqt_amd.seeds import SeedIndex
    from biseqt_amd.sequence import Alphabet
    from oracle import seeds_oracle as SO
    rng = np.random.default_rng(11)
    alphabets = [A, Alphabet('AB'), Alphabet([chr(97 + i) for i in range(20)])]
    for trial in range(60):
        alph = alphabets[trial % 3]
        L = len(alph)
        wordlen = int(rng.integers(1, 9)) if L <= 4 else int(rng.integers(1, 4))
        n, m = int(rng.integers(0, 400)), int(rng.integers(0, 400))
        kind = trial % 5
        s = rng.integers(0, L, n)
        if kind == 0:
            t = rng.integers(0, L, m)
        elif kind == 1:
            t = s.copy()                                     # self comparison by content
        elif kind == 2:
            t = np.concatenate([rng.integers(0, L, 7), s[n // 3:]])
        elif kind == 3:
            s = np.resize(rng.integers(0, L, 3), n); t = np.resize(s[:3], m)     # repeats: many hits per k-mer
        else:
            t = rng.integers(0, L, m); s = s % 2; t = t % 2                      # two letters only
        mask = [set([0]), set([0, 1])] if trial % 4 == 0 else []
        S, T = _seq(alph, s), _seq(alph, t)
        idx = SeedIndex(S, T, wordlen=wordlen, alphabet=alph, mask=mask)
        rows, sc = SO.seed_rows(s, t, wordlen, L, mask)
        assert sc == idx.self_comp
        assert [tuple(r) for r in idx.rows().tolist()] == rows, (trial, wordlen, n, m)
        assert list(idx.seeds()) == SO.seeds(rows, sc)
        assert list(idx.seeds(exclude_trivial=True)) == SO.seeds(rows, sc, exclude_trivial=True)
        for _ in range(4):
            d0, a0 = int(rng.integers(-m - 2, n + 2)), int(rng.integers(0, n + m + 2))
            db, ab = (d0, d0 + int(rng.integers(0, 50))), (a0, a0 + int(rng.integers(0, 200)))
            assert idx.seed_count(d_band=db) == SO.seed_count(rows, d_band=db)
            assert idx.seed_count(a_band=ab) == SO.seed_count(rows, a_band=ab)
            assert idx.seed_count(d_band=db, a_band=ab) == SO.seed_count(rows, db, ab)
            assert list(idx.seeds(d_band=db)) == SO.seeds(rows, sc, d_band=db)
        idx.close()


def test_large_pair_properties(A):
    """1 Mb x 1 Mb, k = 12 (the config-5 shape): too big for the python oracle, so the check is by properties --
    every row is a true k-mer match, the order is (k-mer, i, j) ascending, the count equals the sum over k-mers of
    hits(S) x hits(T) computed with numpy, planted homologies light up their diagonals."""
    from biseqt_amd.seeds import _Index
    rng = np.random.default_rng(5)
    n, k = 1000000, 12
    s = rng.integers(0, 4, n).astype(np.uint8)
    t = rng.integers(0, 4, n).astype(np.uint8)
    t[200000:230000] = s[500000:530000]                       # a planted 30 kb homology on diagonal +300000
    with _Index(s, t, k, A, self_comp=0) as idx:
        nrows = idx.build()
        rows = idx.rows().astype(np.int64)
        planted = idx.count(d_band=(300000, 300000))
    def kmers(x):
        v = np.zeros(len(x) - k + 1, np.int64)
        for q in range(k):
            v = v * 4 + x[q:len(x) - k + 1 + q]
        return v
    ks, kt = kmers(s), kmers(t)
    cs = np.bincount(ks, minlength=4 ** k); ct = np.bincount(kt, minlength=4 ** k)
    assert nrows == int((cs.astype(np.int64) * ct).sum())
    i, j = (rows[:, 1] + rows[:, 0]) // 2, (rows[:, 1] - rows[:, 0]) // 2
    assert (ks[i] == kt[j]).all()
    key = ks[i]
    order = np.lexsort((j, i, key))
    assert (order == np.arange(nrows)).all()
    assert planted >= 30000 - k + 1


def test_gpu_seeds_equal_the_reference_seed_lists(A):
    """The GPU seed enumeration against seed lists and band counts computed by the reference's OWN in-memory enumeration
    (`WordBlotOverlapRef.seeds` / `seed_count`, blot.py:607-642; tests/golden/seed_lists_reference.json.gz, generated by
    importing the reference): the in-memory class row for row in its (j, i) order, `SeedIndex` as the same set."""
    import hashlib
    from biseqt_amd.blot import WordBlotLocalRef, WordBlotOverlapRef
    from biseqt_amd.seeds import SeedIndex
    from tests.helpers import dec, load_golden
    recs = load_golden('seed_lists_reference.json.gz')
    done = 0
    for k, rec in enumerate(recs):
        S, T, w = dec(rec['S']), dec(rec['T']), rec['wordlen']
        if len(S) < w or len(T) < w:
            assert rec['seeds_ij_n'] == 0                   # (sequences shorter than the word: no k-mers at all)
            continue
        # (the product's overlap class refuses self comparisons -- DESIGN.md section 9; its local-similarity sibling shares
        #  the enumeration and takes them)
        cls = WordBlotLocalRef if S == T else WordBlotOverlapRef
        WB = cls(_seq(A, S), wordlen=w, alphabet=A, g_max=0.2, sensitivity=0.9)
        WB._set_query(_seq(A, T))
        rows = [(int(i), int(j)) for i, j in WB.seeds()]
        assert len(rows) == rec['seeds_ij_n'], k
        assert hashlib.sha256(','.join('%d:%d' % r for r in rows).encode()).hexdigest() == rec['seeds_ij_sha256'], k
        for b in rec['band_counts']:
            assert WB.seed_count(d_band=tuple(b['d_band'])) == b['count_d'], (k, b)
            assert WB.seed_count(d_band=tuple(b['d_band']), a_band=tuple(b['a_band'])) == b['count_da'], (k, b)
        idx = SeedIndex(_seq(A, S), _seq(A, T), wordlen=w, alphabet=A)
        assert sorted((int(i), int(j)) for i, j in idx.seeds(exclude_trivial=True)) == sorted(rows), k
        idx.close()
        done += 1
    assert done >= 50
