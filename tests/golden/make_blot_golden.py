"""Generates tests/golden/blot_closed_forms.json and tests/golden/blot_classes.json from the REFERENCE's own Python
(`/root/reference/biseqt/blot.py`), imported in the build container only (the reference never travels: the fixtures
are data -- argument tuples and results, floats as hex).

    python tests/golden/make_blot_golden.py

How the reference (python 2, apsw / SQLite) is made to run under this image's python 3 -- nothing of it is copied or
edited, and no bytecode is written next to it:
  * `apsw` is an empty stand-in module (`biseqt/seeds.py`, `kmers.py` import it at module level; nothing below touches a
    database: the closed forms are module-level functions, the classes used are the SQL-free in-memory variants
    `WordBlotOverlapRef` / `WordBlotLocalRef`, blot.py:582-700);
  * `biseqt.sequence.sha1` is wrapped to encode `str` (python 3 hashes bytes only).  `content_id` is an identity tag,
    not a number that enters any result.
Python-2-only semantics audited in the code these fixtures execute: the only one is the integer division of
`WordBlot.segment_dims` (`K = (a_max - a_min) / 2`, blot.py:300).  Inside `score_seeds` the width is `2 * a_radius`, so
the value is the same in both pythons; in `similar_segments` it decides `scores` for segments of odd width after the
clamp -- those records carry `"scores_py2_safe": false` and their `scores` are not compared.
"""
import hashlib
import json
import os
import sys
import types

import numpy as np

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))


def load_reference():
    sys.dont_write_bytecode = True
    sys.modules['apsw'] = types.ModuleType('apsw')
    sys.path.insert(0, REF)
    import biseqt.sequence as RS
    RS.sha1 = lambda s: hashlib.sha1(s.encode('utf-8') if isinstance(s, str) else s)
    import biseqt.blot as RB
    return RS, RB


def hx(v):
    return float(v).hex()


def closed_forms(RB):
    rng = np.random.default_rng(20261004)
    out = {}
    recs = []
    for _ in range(120):
        n = int(rng.integers(1, 40))
        xs = [float(v) for v in rng.integers(0, 12, n)]
        thr = float(rng.integers(1, 12))
        if rng.random() < 0.5:
            rs = int(rng.integers(0, 6))
        else:
            rs = [int(v) for v in rng.integers(0, 6, n)]
        recs.append({'xs': xs, 'rs': rs, 'threshold': thr, 'peaks': [list(p) for p in RB.find_peaks(xs, rs, thr)]})
    out['find_peaks'] = recs
    recs = []
    for _ in range(300):
        l0, l1 = int(rng.integers(1, 20000)), int(rng.integers(1, 20000))
        d = int(rng.integers(-l1, l0 + 1))
        g = float(rng.choice([0.01, 0.05, 0.1, 0.2, 0.3, 0.5, 0.75])) if rng.random() < 0.7 else float(rng.uniform(0.001, 0.95))
        recs.append({'len0': l0, 'len1': l1, 'diag': d, 'gap_prob': hx(g),
                     'wall_to_wall_distance': int(RB.wall_to_wall_distance(l0, l1, d)),
                     'expected_overlap_len': int(RB.expected_overlap_len(l0, l1, d, g))})
    out['overlap_len'] = recs
    recs = []
    for _ in range(300):
        K = int(rng.integers(0, 200000)) if rng.random() < 0.9 else int(rng.integers(0, 5))
        g = float(rng.choice([0.01, 0.05, 0.1, 0.2, 0.3, 0.5])) if rng.random() < 0.7 else float(rng.uniform(0.001, 0.95))
        s = float(rng.choice([0.9, 0.95, 0.99, 0.999, 1 - 1e-6])) if rng.random() < 0.7 else float(rng.uniform(0.01, 0.9999))
        recs.append({'expected_len': K, 'gap_prob': hx(g), 'sensitivity': hx(s), 'band_radius': int(RB.band_radius(K, g, s))})
    out['band_radius'] = recs
    recs = []
    for _ in range(40):
        Ks = [int(v) for v in rng.integers(0, 50000, int(rng.integers(1, 30)))]
        g, s = float(rng.uniform(0.01, 0.6)), float(rng.uniform(0.5, 0.9999))
        recs.append({'expected_lens': Ks, 'gap_prob': hx(g), 'sensitivity': hx(s),
                     'band_radii': [int(v) for v in RB.band_radii(Ks, g, s)]})
    out['band_radii'] = recs
    recs = []
    for _ in range(300):
        L = int(rng.choice([2, 4, 4, 4, 20]))
        w = int(rng.integers(1, 16))
        area = float(rng.integers(1, 10 ** 8)) if rng.random() < 0.8 else float(rng.uniform(1, 1e9))
        seglen = int(rng.integers(1, 100000))
        p = float(rng.choice([1.0, 0.99, 0.9, 0.8, 0.5])) if rng.random() < 0.5 else float(rng.uniform(0.05, 1.0))
        mu0, sd0 = RB.H0_moments(L, w, area)
        mu1, sd1 = RB.H1_moments(L, w, area, seglen, p)
        recs.append({'alphabet_len': L, 'wordlen': w, 'area': hx(area), 'seglen': seglen, 'p_match': hx(p),
                     'H0': [hx(mu0), hx(sd0)], 'H1': [hx(mu1), hx(sd1)]})
    out['moments'] = recs
    return out


def mutate(rng, s, subst, gap):
    out = []
    for c in s:
        r = rng.random()
        if r < gap / 2:
            continue
        if r < gap:
            out.append(int(rng.integers(0, 4)))
        out.append(int((c + rng.integers(1, 4)) % 4) if rng.random() < subst else int(c))
    return out


def classes(RS, RB):
    rng = np.random.default_rng(20261005)
    A = RS.Alphabet('ACGT')
    out = {'overlap': [], 'local': []}
    # overlaps: suffix of S ~ prefix of T, unrelated pairs, a self comparison
    for case in range(8):
        n, K = int(rng.integers(150, 420)), int(rng.integers(60, 140))
        w = int(rng.choice([4, 5, 6]))
        ov = [int(v) for v in rng.integers(0, 4, K)]
        if case % 4 == 3:
            S = [int(v) for v in rng.integers(0, 4, n)]
            T = [int(v) for v in rng.integers(0, 4, n + 17)]
        elif case == 6:
            S = [int(v) for v in rng.integers(0, 4, n)]
            T = list(S)
        else:
            S = [int(v) for v in rng.integers(0, 4, n - K)] + ov
            T = mutate(rng, ov, 0.05, 0.05) + [int(v) for v in rng.integers(0, 4, n - K)]
        g_max, sens = float(rng.choice([0.1, 0.2, 0.3])), float(rng.choice([0.9, 0.99]))
        WB = RB.WordBlotOverlapRef(RS.Sequence(A, S), wordlen=w, alphabet=A, g_max=g_max, sensitivity=sens)
        Tq = RS.Sequence(A, T)
        scored = WB.score_seeds_(Tq)
        best = WB.highest_scoring_overlap_band(Tq)
        out['overlap'].append({
            'S': ''.join(map(str, S)), 'T': ''.join(map(str, T)), 'wordlen': w, 'g_max': hx(g_max), 'sensitivity': hx(sens),
            'score_seeds': [{'seed': [int(r['seed'][0]), int(r['seed'][1])], 'r': hx(r['r']), 'L': int(r['L']), 'p': hx(r['p'])}
                            for r in scored],
            'best': None if best is None else {'d_band': [hx(best['d_band'][0]), hx(best['d_band'][1])], 'p': hx(best['p']),
                                               'len': int(best['len']), 'score': hx(best['score'])}})
    # local similarities: a planted homology inside unrelated flanks
    for case in range(8):
        n, K = int(rng.integers(200, 480)), int(rng.integers(50, 120))
        w = int(rng.choice([4, 5, 6]))
        hom = [int(v) for v in rng.integers(0, 4, K)]
        a0, b0 = int(rng.integers(0, n - K)), int(rng.integers(0, n - K))
        S = [int(v) for v in rng.integers(0, 4, a0)] + hom + [int(v) for v in rng.integers(0, 4, n - K - a0)]
        T = [int(v) for v in rng.integers(0, 4, b0)] + mutate(rng, hom, 0.06, 0.04) + [int(v) for v in rng.integers(0, 4, n - K - b0)]
        if case == 7:
            T = list(S)
        g_max, sens = float(rng.choice([0.1, 0.2])), float(rng.choice([0.9, 0.99]))
        K_min, p_min = int(rng.choice([30, 50, 80])), float(rng.choice([0.5, 0.7, 0.8]))
        WB = RB.WordBlotLocalRef(RS.Sequence(A, S), wordlen=w, alphabet=A, g_max=g_max, sensitivity=sens)
        Tq = RS.Sequence(A, T)
        scored = WB.score_seeds_(Tq, K_min)
        segs = list(WB.similar_segments(Tq, K_min, p_min, at_least_one=(case % 2 == 0)))
        out['local'].append({
            'S': ''.join(map(str, S)), 'T': ''.join(map(str, T)), 'wordlen': w, 'g_max': hx(g_max), 'sensitivity': hx(sens),
            'K_min': K_min, 'p_min': hx(p_min), 'at_least_one': case % 2 == 0,
            'score_seeds': [{'seed': [int(r['seed'][0]), int(r['seed'][1])], 'neighs': sorted(int(v) for v in r['neighs']),
                             'p': hx(r['p'])} for r in scored],
            'segments': [{'segment': [[int(v) for v in s['segment'][0]], [int(v) for v in s['segment'][1]]], 'p': hx(s['p']),
                          'scores': [hx(s['scores'][0]), hx(s['scores'][1])],
                          'scores_py2_safe': (s['segment'][1][1] - s['segment'][1][0]) % 2 == 0} for s in segs]})
    return out


def main():
    RS, RB = load_reference()
    src = 'generated by tests/golden/make_blot_golden.py from /root/reference/biseqt/blot.py run under python %d.%d ' \
          '(stub apsw, sha1 wrapper: see the script)' % sys.version_info[:2]
    with open(os.path.join(HERE, 'blot_closed_forms.json'), 'w') as f:
        json.dump({'source': src, 'records': closed_forms(RB)}, f)
    with open(os.path.join(HERE, 'blot_classes.json'), 'w') as f:
        json.dump({'source': src, 'records': classes(RS, RB)}, f)
    print('wrote blot_closed_forms.json, blot_classes.json')


if __name__ == '__main__':
    main()
