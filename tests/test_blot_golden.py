"""The Word-Blot closed forms and class-level results against fixtures generated from the REFERENCE's own Python
(tests/golden/make_blot_golden.py runs /root/reference/biseqt/blot.py in the build container: closed forms blot.py:40-218,
in-memory classes blot.py:582-700 on top of :497-579 and :305-490).  Checked here: the CPU oracle (oracle/blot_oracle.py)
and the host-side closed forms of the product (biseqt_amd/blot.py); the device-side classes are checked against the same
fixtures in tests/test_blot_gpu.py."""
import json
import os

import numpy as np
import pytest

from oracle import blot_oracle as BO

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)['records']


def fx(h):
    return float.fromhex(h)


@pytest.mark.parametrize('impl', ['oracle', 'product'])
def test_closed_forms_equal_the_reference(impl):
    if impl == 'oracle':
        M = BO
    else:
        from biseqt_amd import blot as M
    R = _load('blot_closed_forms.json')
    for r in R['find_peaks']:
        assert [list(p) for p in M.find_peaks(r['xs'], r['rs'], r['threshold'])] == r['peaks']
    for r in R['overlap_len']:
        g = fx(r['gap_prob'])
        assert M.wall_to_wall_distance(r['len0'], r['len1'], r['diag']) == r['wall_to_wall_distance']
        assert M.expected_overlap_len(r['len0'], r['len1'], r['diag'], g) == r['expected_overlap_len']
    for r in R['band_radius']:
        assert M.band_radius(r['expected_len'], fx(r['gap_prob']), fx(r['sensitivity'])) == r['band_radius']
    for r in R['band_radii']:
        got = M.band_radii(r['expected_lens'], fx(r['gap_prob']), fx(r['sensitivity']))
        assert [int(v) for v in got] == r['band_radii']
    for r in R['moments']:
        mu0, sd0 = M.H0_moments(r['alphabet_len'], r['wordlen'], fx(r['area']))
        mu1, sd1 = M.H1_moments(r['alphabet_len'], r['wordlen'], fx(r['area']), r['seglen'], fx(r['p_match']))
        assert [float(mu0).hex(), float(sd0).hex()] == r['H0']
        assert [float(mu1).hex(), float(sd1).hex()] == r['H1']
    assert len(R['overlap_len']) == 300 and len(R['moments']) == 300


def _dec(s):
    return [int(c) for c in s]


def test_oracle_overlap_classes_equal_the_reference():
    """WordBlotOverlapRef.score_seeds_ / highest_scoring_overlap_band of the reference, seed for seed."""
    for k, r in enumerate(_load('blot_classes.json')['overlap']):
        S, T = _dec(r['S']), _dec(r['T'])
        g, s = fx(r['g_max']), fx(r['sensitivity'])
        got = BO.score_seeds(S, T, r['wordlen'], 4, g, s, order='mutant')
        assert len(got) == len(r['score_seeds']), k
        for a, b in zip(got, r['score_seeds']):
            assert [int(a['seed'][0]), int(a['seed'][1])] == b['seed'] and float(a['r']).hex() == b['r'], k
            assert int(a['L']) == b['L'] and float(a['p']).hex() == b['p'], k
        best = BO.highest_scoring_overlap_band(S, T, r['wordlen'], 4, g, s, order='mutant')
        if r['best'] is None:
            assert best is None
        else:
            assert [float(best['d_band'][0]).hex(), float(best['d_band'][1]).hex()] == r['best']['d_band'], k
            assert float(best['p']).hex() == r['best']['p'] and int(best['len']) == r['best']['len'], k
            assert float(best['score']).hex() == r['best']['score'], k


def test_oracle_local_classes_equal_the_reference():
    """WordBlotLocalRef.score_seeds_ / similar_segments of the reference: neighbour sets, p per seed, segments in order,
    averaged p (the oracle runs the same KD-tree, so even the order-dependent float sum is identical), z-scores where the
    reference's python-2 integer division cannot differ from what produced the fixture."""
    nseg = 0
    for k, r in enumerate(_load('blot_classes.json')['local']):
        S, T = _dec(r['S']), _dec(r['T'])
        g, s = fx(r['g_max']), fx(r['sensitivity'])
        got = BO.score_seeds_local(S, T, r['wordlen'], 4, g, s, r['K_min'], order='mutant')
        assert len(got) == len(r['score_seeds']), k
        for a, b in zip(got, r['score_seeds']):
            assert [int(a['seed'][0]), int(a['seed'][1])] == b['seed'], k
            assert sorted(int(v) for v in a['neighs']) == b['neighs'] and float(a['p']).hex() == b['p'], k
        segs = BO.similar_segments(S, T, r['wordlen'], 4, g, s, r['K_min'], fx(r['p_min']), at_least_one=r['at_least_one'],
                                   order='mutant')
        assert len(segs) == len(r['segments']), k
        for a, b in zip(segs, r['segments']):
            assert [[int(v) for v in a['segment'][0]], [int(v) for v in a['segment'][1]]] == b['segment'], k
            assert float(a['p']).hex() == b['p'], k
            if b['scores_py2_safe']:
                assert [float(a['scores'][0]).hex(), float(a['scores'][1]).hex()] == b['scores'], k
            nseg += 1
    assert nseg >= 6
