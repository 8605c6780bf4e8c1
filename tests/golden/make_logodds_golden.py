"""Generates tests/golden/logodds_reference.json and tests/golden/seed_lists_reference.json.gz from the REFERENCE's own Python,
imported in the build container only (the reference never travels: the fixtures are data -- arguments and results, floats
as hex, sequences as digit strings).

    python tests/golden/make_logodds_golden.py

* `MutationProcess(...).log_odds_scores()` (`/root/reference/biseqt/stochastics.py:234-310`) on a grid of substitution
  probabilities, gap probabilities, alphabets of 2 / 4 / 20 letters and null hypotheses,
  including the five noise levels of the reference's `tests/test_pw.py:106` (subst = go = ge = err).  These are the float
  scores the f64 DP path is fed with: `tests/golden/make_golden.py` takes the scores of `float_logodds.json` from the same
  function, and `tests/test_host.py` holds `biseqt_amd.stochastics` to the fixture bit for bit.
* the seed lists of the in-memory enumeration (`WordBlotOverlapRef.seeds`, `/root/reference/biseqt/blot.py:607-625`: for
  every position of T, the hits of its k-mer in S, in (j, i) order) for seeded pairs -- related, unrelated, repeats,
  low-complexity, self comparisons with and without the trivial seeds -- which pin `oracle/seeds_oracle.py` and the GPU
  seed enumeration (`biseqt_amd/seeds.py`) row for row.

`biseqt.stochastics` imports under python 3 as it is; `blot.py` needs the stand-ins documented in make_blot_golden.py
(an empty `apsw` module; `sha1` of a str).  Nothing of the reference is copied, edited or written next to it.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_blot_golden import load_reference, hx, mutate      # noqa: E402


def logodds(RS):
    import biseqt.stochastics as RT
    rng = np.random.default_rng(20261006)
    recs = []

    def add(letters, subst_probs, go, ge, null=None, insert_dist=None, tag=''):
        A = RS.Alphabet(letters)
        M = RT.MutationProcess(A, subst_probs=subst_probs, go_prob=go, ge_prob=ge, insert_dist=insert_dist)
        S, (gos, ges) = M.log_odds_scores(null_hypothesis=null) if null is not None else M.log_odds_scores()
        recs.append({'tag': tag, 'letters': list(letters),
                     'subst_probs': hx(subst_probs) if not isinstance(subst_probs, list) else [[hx(v) for v in row] for row in subst_probs],
                     'go_prob': hx(go), 'ge_prob': hx(ge), 'null': None if null is None else [hx(v) for v in null],
                     'subst_scores': [[hx(v) for v in row] for row in S], 'go_score': hx(gos), 'ge_score': hx(ges)})

    for err in (1e-2, 1e-1, 2e-1, 3e-1, 4e-1):                   # tests/test_pw.py:106-140, 157-185
        add('ACGT', err, err, err, tag='test_pw noise level')
        add('ACGT', err, err / 2, err, tag='affine variant (go_prob < ge_prob)')
    for letters in ('AC', 'ACGT', 'ACDEFGHIKLMNPQRSTVWY'):
        for _ in range(12):
            s = float(rng.choice([0.01, 0.05, 0.1, 0.25, 0.4])) if rng.random() < 0.6 else float(rng.uniform(0.005, 0.6))
            ge = float(rng.choice([0.05, 0.1, 0.3, 0.5])) if rng.random() < 0.6 else float(rng.uniform(0.01, 0.8))
            go = ge * float(rng.choice([1.0, 0.5, 0.1, 0.01]))
            add(letters, s, go, ge, tag='scalar substitution probability')
    # non-uniform null hypotheses.  (A full MATRIX of substitution probabilities cannot be recorded: the reference's
    # constructor never stores a list it is given -- stochastics.py:127-134 assigns self.subst_probs only in the scalar
    # branch -- so log_odds_scores raises AttributeError for it.)
    for _ in range(16):
        L = int(rng.choice([2, 4, 20]))
        null = rng.uniform(0.1, 1.0, L)
        null = [float(v) for v in (null / null.sum())]
        s = float(rng.uniform(0.005, 0.6))
        ge = float(rng.uniform(0.05, 0.6))
        add('ACDEFGHIKLMNPQRSTVWY'[:L], s, ge * float(rng.uniform(0.05, 1.0)), ge, null=null, tag='own null hypothesis')
    return recs


def rows_record(key, rows):
    """A row list as data: its length and the SHA-256 of "i:j,i:j,..." always, the rows themselves up to 1500 of them."""
    import hashlib
    rows = [(int(i), int(j)) for (i, j) in rows]
    out = {key + '_n': len(rows), key + '_sha256': hashlib.sha256(','.join('%d:%d' % r for r in rows).encode()).hexdigest()}
    if len(rows) <= 1500:
        out[key] = [list(r) for r in rows]
    return out


def seed_lists(RS, RB):
    rng = np.random.default_rng(20261007)
    A = RS.Alphabet('ACGT')
    recs = []
    for case in range(64):
        kind = case % 8
        w = int(rng.choice([3, 4, 5, 6, 8]))
        n = int(rng.integers(40, 400))
        S = [int(v) for v in rng.integers(0, 4, n)]
        if kind == 0:                                             # related
            T = mutate(rng, S, 0.05, 0.05)
        elif kind == 1:                                           # unrelated
            T = [int(v) for v in rng.integers(0, 4, int(rng.integers(40, 400)))]
        elif kind == 2:                                           # a repeat unit, many hits per k-mer
            unit = [int(v) for v in rng.integers(0, 4, int(rng.integers(2, 9)))]
            S = (unit * (n // len(unit) + 1))[:n]
            T = mutate(rng, S, 0.03, 0.02)
        elif kind == 3:                                           # low complexity: two letters
            S = [int(v) for v in rng.integers(0, 2, n)]
            T = [int(v) for v in rng.integers(0, 2, int(rng.integers(30, 200)))]
        elif kind == 4:                                           # self comparison (same content): trivial seeds excluded
            T = list(S)
        elif kind == 5:                                           # suffix / prefix overlap
            k = int(rng.integers(10, n))
            T = S[k:] + [int(v) for v in rng.integers(0, 4, int(rng.integers(0, 120)))]
        elif kind == 6:                                           # shorter than the word, or barely longer
            S = S[:int(rng.integers(0, w + 3))]
            T = [int(v) for v in rng.integers(0, 4, int(rng.integers(0, w + 3)))]
        else:                                                     # self comparison of a repeat
            unit = [int(v) for v in rng.integers(0, 4, int(rng.integers(3, 7)))]
            S = (unit * (n // len(unit) + 1))[:n]
            T = list(S)
        WB = RB.WordBlotOverlapRef(RS.Sequence(A, S), wordlen=w, alphabet=A, g_max=0.2, sensitivity=0.9)
        WB.T = RS.Sequence(A, T)
        rec = {'kind': kind, 'S': ''.join(map(str, S)), 'T': ''.join(map(str, T)), 'wordlen': w}
        rec.update(rows_record('seeds_ij', WB.seeds()))
        if S == T:
            WB._seeds = {}
            rec.update(rows_record('seeds_ij_with_trivial', WB.seeds(exclude_trivial=False)))
        # band counts of the same rows (seed_count, blot.py:627-642) for a few bands
        bands = []
        for _ in range(4):
            d0 = int(rng.integers(-len(T) - 2, len(S) + 2)); d1 = d0 + int(rng.integers(0, 60))
            a0 = int(rng.integers(0, len(S) + len(T) + 2)); a1 = a0 + int(rng.integers(0, 200))
            WB._seeds = {}
            bands.append({'d_band': [d0, d1], 'a_band': [a0, a1], 'count_d': int(WB.seed_count(d_band=(d0, d1))),
                          'count_da': int(WB.seed_count(d_band=(d0, d1), a_band=(a0, a1)))})
        rec['band_counts'] = bands
        recs.append(rec)
    return recs


def main():
    # (stochastics first: blot.py turns every warning into an error at import, and stochastics.py's docstrings carry
    #  escape sequences python 3 warns about when it compiles them)
    sys.dont_write_bytecode = True
    sys.path.insert(0, '/root/reference')
    import biseqt.stochastics  # noqa: F401
    RS, RB = load_reference()
    src = 'generated by tests/golden/make_logodds_golden.py from /root/reference/biseqt/{stochastics,blot}.py run under python %d.%d ' \
          '(blot.py with the stand-ins of make_blot_golden.py)' % sys.version_info[:2]
    with open(os.path.join(HERE, 'logodds_reference.json'), 'w') as f:
        json.dump({'source': src, 'records': logodds(RS)}, f)
    import gzip
    with gzip.GzipFile(os.path.join(HERE, 'seed_lists_reference.json.gz'), 'wb', mtime=0) as f:
        f.write(json.dumps({'source': src, 'records': seed_lists(RS, RB)}).encode())
    print('wrote logodds_reference.json, seed_lists_reference.json.gz')


if __name__ == '__main__':
    main()
