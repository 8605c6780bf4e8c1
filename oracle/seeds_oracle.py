"""CPU restatement of the reference's exact-match seed enumeration (TEST INFRASTRUCTURE, never the product path).

Follows, function by function (reference = /root/reference/biseqt):
  * kmer_as_int        kmers.py:164-210   k-mer -> integer, letters as digits in base |alphabet|
  * as_kmer_seq        kmers.py:213-241   one integer per position, None where the letter SET equals a mask set
  * seed_rows          seeds.py:117-162   rows (d, a) of the seeds table, in insertion (rowid) order:
                                          k-mers ascending (kmers.py:497-509: DISTINCT over the SQL index), hits
                                          of a k-mer in (seqid, pos) order (kmers.py:480-495), then
                                          combinations(hits, 2) restricted to different sequences -- or, for a
                                          self comparison (S == T by content, seeds.py:33), all combinations
                                          followed by the trivial pairs (x, x)
  * seeds              seeds.py:164-197   (i, j) per row, optionally in a diagonal band; self comparisons also
                                          yield the mirror (j, i), trivial seeds can be excluded
  * seed_count         seeds.py:199-237   COUNT(*) of rows in a d band and / or an a band
  * coordinate maps    seeds.py:55-106    (python-2 integer division: a + d and a - d are always even)
  * seeds_by_mutant    blot.py:607-620    the in-memory variant: for pos in T, for hit in S -> (hit, pos)

Parity pin: the reference's Python cannot be imported here (python 2 + apsw/SQLite), so this
restatement is pinned by the known answers of the reference's own tests (tests/test_seeds.py, tests/test_kmers.py,
the seeds.py docstring), restated in tests/test_seeds_oracle.py.
"""
from itertools import combinations, product


def kmer_as_int(contents, L):
    v = 0
    for c in contents:
        assert 0 <= c < L
        v = v * L + int(c)
    return v


def as_kmer_seq(contents, wordlen, L, mask=()):
    out = []
    mask = [set(m) for m in mask]
    for pos in range(len(contents) - wordlen + 1):
        word = contents[pos: pos + wordlen]
        if mask and set(int(c) for c in word) in mask:
            out.append(None)
            continue
        out.append(kmer_as_int(word, L))
    return out


def to_diagonal_coordinates(i, j):
    return i - j, i + j


def to_ij_coordinates(d, a):
    return (a + d) // 2, (a - d) // 2


def to_ij_coordinates_seg(seg):
    corners = [to_ij_coordinates(d, a) for d, a in product(*seg)]
    i_start = max(min(i for i, _ in corners), 0)
    j_start = max(min(j for _, j in corners), 0)
    return (i_start, max(i for i, _ in corners)), (j_start, max(j for _, j in corners))


def seed_rows(S, T, wordlen, L, mask=(), self_comp=None):
    """Rows (d, a) of the reference's seeds table in rowid order."""
    S, T = [int(c) for c in S], [int(c) for c in T]
    if self_comp is None:
        self_comp = S == T
    hits = {}
    for pos, k in enumerate(as_kmer_seq(S, wordlen, L, mask)):
        if k is not None:
            hits.setdefault(k, []).append((1, pos))
    if not self_comp:
        for pos, k in enumerate(as_kmer_seq(T, wordlen, L, mask)):
            if k is not None:
                hits.setdefault(k, []).append((2, pos))
    rows = []
    for k in sorted(hits):
        h = hits[k]
        if self_comp:
            pairs = list(combinations(h, 2)) + [(x, x) for x in h]
        else:
            pairs = [(p, q) for p, q in combinations(h, 2) if p[0] != q[0]]
        for (_, pos0), (_, pos1) in pairs:
            rows.append(to_diagonal_coordinates(pos0, pos1))
    return rows, self_comp


def seeds(rows, self_comp, d_band=None, exclude_trivial=False):
    out = []
    for d, a in rows:
        if d_band is not None and not (d_band[0] <= d <= d_band[1]):
            continue
        i, j = to_ij_coordinates(d, a)
        if self_comp and exclude_trivial and i == j:
            continue
        out.append((i, j))
        if self_comp and i != j:
            out.append((j, i))
    return out


def seed_count(rows, d_band=None, a_band=None):
    n = 0
    for d, a in rows:
        if d_band is not None and not (d_band[0] <= d <= d_band[1]):
            continue
        if a_band is not None and not (a_band[0] <= a <= a_band[1]):
            continue
        n += 1
    return n


def seeds_by_mutant(S, T, wordlen, L, exclude_trivial=True):
    """blot.py:607-620: hits of the reference sequence per k-mer, then T scanned left to right."""
    S, T = [int(c) for c in S], [int(c) for c in T]
    table = {}
    for pos, k in enumerate(as_kmer_seq(S, wordlen, L)):
        table.setdefault(k, []).append(pos)
    same = S == T
    out = []
    for pos, k in enumerate(as_kmer_seq(T, wordlen, L)):
        for pos_ref in table.get(k, ()):
            if same and exclude_trivial and pos == pos_ref:
                continue
            out.append((pos_ref, pos))
    return out
