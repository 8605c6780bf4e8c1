"""The band-selection oracle against what the reference's tests/test_blot.py:16-114,160-197 assert (restated) and
the two values SURVEY 8f records from the reference itself."""
import numpy as np
import pytest

from biseqt_amd import synth
from oracle import blot_oracle as BO


def test_recorded_reference_values():
    assert BO.band_radius(2000, .2, .99) == 52
    assert BO.expected_overlap_len(5000, 5000, 1000, .2) == 4445


def test_find_peaks():                              # tests/test_blot.py:16-32
    assert BO.find_peaks([0, 1, 2, 3, 100, 5, 6, 7, 8], 3, 100) == [(3, 5)]
    assert BO.find_peaks([0, 1, 2, 3, 100, 5, 6, 7, 8, 9, 10, 11, 100, 13, 14, 15, 16], 3, 100) == [(3, 5), (11, 13)]
    assert BO.find_peaks([0, 1, 2, 3, 100, 5, 6, 7, 100, 9, 10, 11, 12, 13], 3, 100) == [(3, 9)]


def test_expected_overlap_len():                    # tests/test_blot.py:35-61
    n, gap = 50, .1
    lens = np.array([BO.expected_overlap_len(n, n, d, gap) for d in range(-n, n + 1)])
    assert np.all(np.diff(lens[0:n]) >= 0) and lens[n] >= n and np.all(np.diff(lens[n:]) <= 0)
    gaps = [i * .05 for i in range(6)]
    assert np.all(np.diff([BO.expected_overlap_len(n, n, 0, g) for g in gaps]) >= 0)


def test_band_radius_and_radii():                   # tests/test_blot.py:64-112
    Ks = [i * 200 for i in range(1, 10)]
    Rs = [BO.band_radius(K, .1, 1 - 1e-3) for K in Ks]
    ratios = np.array([Rs[i] / np.sqrt(Ks[i]) for i in range(len(Ks))])
    assert np.allclose(ratios - ratios[0], 0, atol=1e-1)
    assert np.all(np.diff([BO.band_radius(50, i * .05, .99) for i in range(1, 7)]) >= 0)
    assert np.all(np.diff([BO.band_radius(50, .1, 1 - i * .05) for i in range(1, 7)]) <= 0)
    radii = BO.band_radii(range(50), gap_prob=.1, sensitivity=.99)
    assert len(radii) == 50 and np.all(np.diff(radii) >= 0)
    lens = [BO.expected_overlap_len(50, 50, d, .1) for d in range(-50, 50)]
    radii = BO.band_radii(lens, .1, .99)
    assert len(radii) == 100 and np.all(np.diff(radii[0:50]) >= 0) and np.all(np.diff(radii[50:]) <= 0)
    assert all(BO.band_radius(K, .1, .99) == r for K, r in zip(lens, radii))


@pytest.mark.parametrize('wordlen,K,n', [(8, 500, 2000), (8, 1000, 2000), (15, 500, 2000)])
def test_overlap_detection(wordlen, K, n):          # tests/test_blot.py:160-197
    gap, subst = .05, .05
    rng = synth.rng_for(wordlen * 1000 + K)
    p_match = (1 - gap) * (1 - subst)
    overlap = synth.rand_seqs(rng, 1, K)[0]
    S = np.concatenate([synth.rand_seqs(rng, 1, n - K)[0], overlap])
    T = np.concatenate([synth.mutate(rng, overlap, subst, gap, gap), synth.rand_seqs(rng, 1, n - K)[0]])
    rec = BO.highest_scoring_overlap_band(S.tolist(), T.tolist(), wordlen, 4, .2, .99)
    d_min, d_max = rec['d_band']
    assert d_min * .9 < n - K < 1.1 * d_max
    assert rec['p'] > .9 * p_match
    S = np.concatenate([synth.rand_seqs(rng, 1, n - K)[0], overlap, synth.rand_seqs(rng, 1, n)[0]])
    T = np.concatenate([synth.rand_seqs(rng, 1, n)[0], synth.mutate(rng, overlap, subst, gap, gap), synth.rand_seqs(rng, 1, n - K)[0]])
    rec = BO.highest_scoring_overlap_band(S.tolist(), T.tolist(), wordlen, 4, .2, .99)
    assert rec['p'] < p_match


@pytest.mark.parametrize('wordlen,K,n', [(8, 500, 2000), (8, 1000, 2000), (15, 500, 2000)])
def test_local_similarity(wordlen, K, n):           # tests/test_blot.py:117-152
    gap, subst = .05, .05
    rng = synth.rng_for(wordlen * 100 + K)
    hom = synth.rand_seqs(rng, 1, K)[0]
    S = np.concatenate([hom, synth.rand_seqs(rng, 1, n - K)[0]])
    T = np.concatenate([synth.mutate(rng, hom, subst, gap, gap), synth.rand_seqs(rng, 1, n - K)[0]])
    p_match = (1 - gap) * (1 - subst) * .9
    homs = BO.similar_segments(S.tolist(), T.tolist(), wordlen, 4, .2, .99, K, p_match)
    assert len(homs) == 1
    (d_min, d_max), (a_min, a_max) = homs[0]['segment']
    assert d_min < 10 and d_max > -10 and a_min < K
    assert 0.8 * p_match <= homs[0]['p'] <= 1.2 * p_match
