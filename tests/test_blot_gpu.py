"""GPU band selection (biseqt_amd.blot.WordBlotOverlap over include/pw_seeds.h) against the oracle, which runs the
reference's own neighbour search (scipy cKDTree) on the CPU; and the reference's overlap-detection test
(tests/test_blot.py:160-197) end to end: seeds -> band -> banded overlap alignment on the GPU."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mk(A, arr):
    from biseqt_amd.sequence import Sequence
    return Sequence(A, tuple(int(c) for c in arr))


def test_closed_forms_match_oracle():
    from biseqt_amd import blot as B
    from oracle import blot_oracle as BO
    assert B.band_radius(2000, .2, .99) == 52 and B.expected_overlap_len(5000, 5000, 1000, .2) == 4445
    for g, s in ((.1, .99), (.2, .999), (.05, .9)):
        lens = [B.expected_overlap_len(700, 900, d, g) for d in range(-900, 701)]
        assert lens == [BO.expected_overlap_len(700, 900, d, g) for d in range(-900, 701)]
        assert B.band_radii(lens, g, s).tolist() == BO.band_radii(lens, g, s).tolist()
        assert all(B.band_radius(K, g, s) == BO.band_radius(K, g, s) for K in lens[::37])
    assert B.find_peaks([0, 1, 2, 3, 100, 5, 6, 7, 100, 9, 10, 11, 12, 13], 3, 100) == [(3, 9)]
    for args in ((4, 8, 1234.), (4, 15, 1e6), (20, 3, 50.)):
        assert B.H0_moments(*args) == BO.H0_moments(*args)
        assert B.H1_moments(*args, 500, .9) == BO.H1_moments(*args, 500, .9)


@pytest.mark.parametrize('wordlen,K,n', [(8, 500, 2000), (8, 1000, 2000), (15, 500, 2000), (6, 300, 1200)])
def test_score_seeds_and_best_band_vs_oracle(wordlen, K, n):
    from biseqt_amd import synth
    from biseqt_amd.blot import WordBlotOverlap
    from biseqt_amd.sequence import Alphabet
    from oracle import blot_oracle as BO
    A = Alphabet('ACGT')
    rng = synth.rng_for(wordlen * 1000 + K)
    overlap = synth.rand_seqs(rng, 1, K)[0]
    S = np.concatenate([synth.rand_seqs(rng, 1, n - K)[0], overlap])
    T = np.concatenate([synth.mutate(rng, overlap, .05, .05, .05), synth.rand_seqs(rng, 1, n - K)[0]])
    for (s, t) in ((S, T), (T, S), (S[:300], T[:40])):
        wb = WordBlotOverlap(_mk(A, s), _mk(A, t), g_max=.2, sensitivity=.99, alphabet=A, wordlen=wordlen)
        got = wb.score_seeds()
        exp = BO.score_seeds(s.tolist(), t.tolist(), wordlen, 4, .2, .99)
        assert len(got) == len(exp)
        for g, e in zip(got, exp):
            assert g['seed'] == e['seed'] and g['r'] == e['r'] and g['L'] == e['L'] and g['p'] == e['p'], (g, e)
        rec = wb.highest_scoring_overlap_band()
        ref = BO.highest_scoring_overlap_band(s.tolist(), t.tolist(), wordlen, 4, .2, .99)
        assert (rec is None) == (ref is None)
        if rec is not None:
            assert rec['d_band'] == ref['d_band'] and rec['p'] == ref['p'] and rec['len'] == ref['len']
            assert rec['score'] == ref['score']
        wb.close()


def test_overlap_detection_then_banded_alignment():
    """tests/test_blot.py:160-197 restated, then the band goes to the banded overlap aligner."""
    from biseqt_amd import synth
    from biseqt_amd.blot import WordBlotOverlap
    from biseqt_amd.pw import Aligner, BANDED_MODE, B_OVERLAP
    from biseqt_amd.sequence import Alphabet
    A = Alphabet('ACGT')
    gap, subst, n, K = .05, .05, 5000, 1000
    rng = synth.rng_for(77)
    overlap = synth.rand_seqs(rng, 1, K)[0]
    S = _mk(A, np.concatenate([synth.rand_seqs(rng, 1, n - K)[0], overlap]))
    T = _mk(A, np.concatenate([synth.mutate(rng, overlap, subst, gap, gap), synth.rand_seqs(rng, 1, n - K)[0]]))
    wb = WordBlotOverlap(S, T, g_max=.2, sensitivity=.99, alphabet=A, wordlen=8)
    rec = wb.highest_scoring_overlap_band()
    d_min, d_max = rec['d_band']
    assert d_min * .9 < n - K < 1.1 * d_max and rec['p'] > .9 * (1 - gap) * (1 - subst)
    band = (max(int(d_min), -len(T)), min(int(d_max), len(S)))
    with Aligner(S, T, alnmode=BANDED_MODE, alntype=B_OVERLAP, diag_range=band,
                 match_score=1, mismatch_score=-3, go_score=-5, ge_score=-2) as aligner:
        score = aligner.solve()
        aln = aligner.traceback()
    assert score is not None and score > 0.5 * K
    assert abs(aln.origin_start - (n - K)) < 30 and aln.mutant_start == 0


@pytest.mark.parametrize('wordlen,K,n', [(8, 500, 2000), (8, 1000, 2000), (15, 500, 2000), (6, 200, 900)])
def test_local_similarity_vs_oracle(wordlen, K, n):
    """WordBlot.score_seeds / similar_segments (blot.py:376-490) against the oracle (which runs the reference's
    KD-tree search and depth-first growth on the CPU): neighbour sets, p per seed (==), segments (==), seed counts;
    the segment's averaged p within 1e-12 relative -- its last bits follow the KD-tree's internal neighbour order
    in the reference -- and the z-scores derived from it within 1e-9."""
    from biseqt_amd import synth
    from biseqt_amd.blot import WordBlot
    from biseqt_amd.sequence import Alphabet
    from oracle import blot_oracle as BO
    A = Alphabet('ACGT')
    gap, subst = .05, .05
    rng = synth.rng_for(wordlen * 100 + K)
    hom = synth.rand_seqs(rng, 1, K)[0]
    S = np.concatenate([hom, synth.rand_seqs(rng, 1, n - K)[0]])
    T = np.concatenate([synth.mutate(rng, hom, subst, gap, gap), synth.rand_seqs(rng, 1, n - K)[0]])
    # a second, weaker homology elsewhere so that several segments and thresholds are exercised
    seg2 = synth.mutate(rng, S[n // 2:n // 2 + 200], .1, .05, .05)[:180]
    T[n - 300:n - 300 + len(seg2)] = seg2
    p_match = (1 - gap) * (1 - subst) * .9
    wb = WordBlot(_mk(A, S), _mk(A, T), g_max=.2, sensitivity=.99, alphabet=A, wordlen=wordlen)
    for Kq in (K, max(K // 4, 40)):
        got = wb.score_seeds(Kq)
        exp = BO.score_seeds_local(S.tolist(), T.tolist(), wordlen, 4, .2, .99, Kq)
        assert len(got) == len(exp)
        for g, e in zip(got, exp):
            assert g['seed'] == e['seed'] and sorted(g['neighs']) == sorted(e['neighs']) and g['p'] == e['p']
        for p_min in (p_match, .5, .99):
            gs = list(wb.similar_segments(Kq, p_min))
            es = BO.similar_segments(S.tolist(), T.tolist(), wordlen, 4, .2, .99, Kq, p_min)
            assert [g['segment'] for g in gs] == [e['segment'] for e in es], (Kq, p_min)
            for g, e in zip(gs, es):
                assert abs(g['p'] - e['p']) <= 1e-12 * max(abs(e['p']), 1e-300)
                assert np.allclose(g['scores'], e['scores'], rtol=1e-9, atol=0)
    gs = list(wb.similar_segments(K, 1.5, at_least_one=True))
    es = BO.similar_segments(S.tolist(), T.tolist(), wordlen, 4, .2, .99, K, 1.5, at_least_one=True)
    assert len(gs) == len(es) == 1 and gs[0]['segment'] == es[0]['segment']
    # the reference's own assertions (tests/test_blot.py:117-152) on the main homology
    homs = list(wb.similar_segments(K, p_match))
    assert any(h['segment'][0][0] < 10 and h['segment'][0][1] > -10 and h['segment'][1][0] < K
               and 0.8 * p_match <= h['p'] <= 1.2 * p_match for h in homs)
    wb.close()


def test_adversarial_fuzz_of_seeds_bands_segments():
    """A bounded run of tests/micro/fuzz_seeds_gpu.py (3366 cases clean in the 2-minute run recorded in DESIGN.md):
    rows and their order, band counts, the best overlap band (single-pair and batched paths), similar segments --
    repeats, two-letter sequences, shared blocks on several diagonals, all against the oracles."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location('fuzz_seeds_gpu', os.path.join(os.path.dirname(__file__), 'micro', 'fuzz_seeds_gpu.py'))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    n, bad = fz.run(12, 20261004)
    assert bad == 0 and n > 50


@pytest.mark.parametrize('wordlen', [8, 10])
def test_in_memory_ref_variants(wordlen):
    """WordBlotLocalRef / WordBlotOverlapRef (blot.py:582-700): one reference sequence, many queries, seeds iterated
    with T scanned left to right -- against the oracle in that order; and the MemoryError of long words
    (tests/test_blot.py:134-136, 175-177)."""
    from biseqt_amd import synth
    from biseqt_amd.blot import WordBlotLocalRef, WordBlotOverlapRef
    from biseqt_amd.sequence import Alphabet
    from oracle import blot_oracle as BO, seeds_oracle as SO
    A = Alphabet('ACGT')
    kw = dict(g_max=.2, sensitivity=.99, alphabet=A, wordlen=wordlen)
    rng = synth.rng_for(wordlen)
    n, K = 2000, 500
    hom = synth.rand_seqs(rng, 1, K)[0]
    s = np.concatenate([synth.rand_seqs(rng, 1, n - K)[0], hom])
    queries = [np.concatenate([synth.mutate(rng, hom, .05, .05, .05), synth.rand_seqs(rng, 1, n - K)[0]]),
               np.concatenate([synth.rand_seqs(rng, 1, 700)[0], synth.mutate(rng, s[300:900], .03, .02, .2)]),
               synth.rand_seqs(rng, 1, 500)[0]]
    with pytest.raises(MemoryError):
        WordBlotLocalRef(_mk(A, s), allowed_memory=1, **dict(kw, wordlen=15))
    with pytest.raises(MemoryError):
        WordBlotOverlapRef(_mk(A, s), allowed_memory=1, **dict(kw, wordlen=15))
    loc = WordBlotLocalRef(_mk(A, s), allowed_memory=1, **kw)
    ovl = WordBlotOverlapRef(_mk(A, s), allowed_memory=1, **kw)
    for t in queries:
        T = _mk(A, t)
        got = list(loc.similar_segments(T, 300, .7))
        exp = BO.similar_segments(s.tolist(), t.tolist(), wordlen, 4, .2, .99, 300, .7, order='mutant')
        assert [g['segment'] for g in got] == [e['segment'] for e in exp]
        for g, e in zip(got, exp):
            assert abs(g['p'] - e['p']) <= 1e-12 * max(abs(e['p']), 1e-300)
        assert loc.seeds() == SO.seeds_by_mutant(s.tolist(), t.tolist(), wordlen, 4)
        sc = loc.score_seeds_(T, 300)
        ex = BO.score_seeds_local(s.tolist(), t.tolist(), wordlen, 4, .2, .99, 300, order='mutant')
        assert [(x['seed'], x['p'], sorted(x['neighs'])) for x in sc] == [(x['seed'], x['p'], sorted(x['neighs'])) for x in ex]
        rec = ovl.highest_scoring_overlap_band(T)
        ref = BO.highest_scoring_overlap_band(s.tolist(), t.tolist(), wordlen, 4, .2, .99, order='mutant')
        assert (rec is None) == (ref is None)
        if rec is not None:
            assert rec['d_band'] == ref['d_band'] and rec['p'] == ref['p'] and rec['score'] == ref['score']
    loc.close(); ovl.close()


def test_self_similarity_vs_oracle():
    """WordBlot(S, S): the reference turns equal contents into a self comparison (seeds.py:33) -- the seeds it iterates
    are the non-trivial matches, each followed by its mirror image, while seed_count still counts table rows.  A
    sequence with internal repeats against itself: score_seeds and similar_segments equal to the oracle's."""
    from biseqt_amd import synth
    from biseqt_amd.blot import WordBlot
    from biseqt_amd.sequence import Alphabet
    from oracle import blot_oracle as BO
    A = Alphabet('ACGT')
    rng = synth.rng_for(808)
    unit = synth.rand_seqs(rng, 1, 300)[0]
    s = np.concatenate([synth.rand_seqs(rng, 1, 400)[0], unit, synth.rand_seqs(rng, 1, 350)[0],
                        synth.mutate(rng, unit, .04, .02, .3), synth.rand_seqs(rng, 1, 200)[0]])
    for wordlen, K in ((6, 120), (8, 200)):
        wb = WordBlot(_mk(A, s), _mk(A, s), g_max=.2, sensitivity=.99, alphabet=A, wordlen=wordlen)
        assert wb.self_comp
        got = wb.score_seeds(K)
        exp = BO.score_seeds_local(s.tolist(), s.tolist(), wordlen, 4, .2, .99, K)
        assert [(g['seed'], g['p'], sorted(g['neighs'])) for g in got] == [(e['seed'], e['p'], sorted(e['neighs'])) for e in exp]
        for p_min in (.6, .85):
            gs = list(wb.similar_segments(K, p_min))
            es = BO.similar_segments(s.tolist(), s.tolist(), wordlen, 4, .2, .99, K, p_min)
            assert [g['segment'] for g in gs] == [e['segment'] for e in es] and len(gs) >= 2
            for g, e in zip(gs, es):
                assert abs(g['p'] - e['p']) <= 1e-12 * max(abs(e['p']), 1e-300)
                assert np.allclose(g['scores'], e['scores'], rtol=1e-9, atol=0)
        wb.close()


def _fx(h):
    return float.fromhex(h)


def test_ref_classes_equal_the_reference_fixtures():
    """WordBlotOverlapRef / WordBlotLocalRef of the product against results of the REFERENCE's own classes
    (tests/golden/blot_classes.json, generated by make_blot_golden.py from /root/reference/biseqt/blot.py): every seed's
    r / L / p and neighbour set, the best overlap band with its z-score, the local segments in order with their averaged
    p (1e-12: the reference's float sum follows its KD-tree's neighbour order) and z-scores (1e-9, where the reference's
    python-2 integer division cannot have differed)."""
    import json
    from biseqt_amd.blot import WordBlotLocalRef, WordBlotOverlapRef
    from biseqt_amd.sequence import Alphabet
    A = Alphabet('ACGT')
    with open(os.path.join(os.path.dirname(__file__), 'golden', 'blot_classes.json')) as f:
        R = json.load(f)['records']
    for k, r in enumerate(R['overlap']):
        S, T = [int(c) for c in r['S']], [int(c) for c in r['T']]
        if S == T:
            continue           # documented divergence: the overlap classes refuse to compare a sequence with itself
        ovl = WordBlotOverlapRef(_mk(A, np.array(S)), wordlen=r['wordlen'], alphabet=A, g_max=_fx(r['g_max']),
                                 sensitivity=_fx(r['sensitivity']))
        Tq = _mk(A, np.array(T))
        got = ovl.score_seeds_(Tq)
        assert len(got) == len(r['score_seeds']), k
        for a, b in zip(got, r['score_seeds']):
            assert [a['seed'][0], a['seed'][1]] == b['seed'] and float(a['r']).hex() == b['r'], k
            assert a['L'] == b['L'] and float(a['p']).hex() == b['p'], k
        best = ovl.highest_scoring_overlap_band(Tq)
        if r['best'] is None:
            assert best is None
        else:
            assert [float(best['d_band'][0]).hex(), float(best['d_band'][1]).hex()] == r['best']['d_band'], k
            assert float(best['p']).hex() == r['best']['p'] and best['len'] == r['best']['len'], k
            assert float(best['score']).hex() == r['best']['score'], k
        ovl.close()
    nseg = 0
    for k, r in enumerate(R['local']):
        S, T = [int(c) for c in r['S']], [int(c) for c in r['T']]
        loc = WordBlotLocalRef(_mk(A, np.array(S)), wordlen=r['wordlen'], alphabet=A, g_max=_fx(r['g_max']),
                               sensitivity=_fx(r['sensitivity']))
        Tq = _mk(A, np.array(T))
        got = loc.score_seeds_(Tq, r['K_min'])
        assert len(got) == len(r['score_seeds']), k
        for a, b in zip(got, r['score_seeds']):
            assert [a['seed'][0], a['seed'][1]] == b['seed'], k
            assert sorted(int(v) for v in a['neighs']) == b['neighs'] and float(a['p']).hex() == b['p'], k
        segs = list(loc.similar_segments(Tq, r['K_min'], _fx(r['p_min']), at_least_one=r['at_least_one']))
        assert len(segs) == len(r['segments']), k
        for a, b in zip(segs, r['segments']):
            assert [list(a['segment'][0]), list(a['segment'][1])] == b['segment'], k
            assert abs(a['p'] - _fx(b['p'])) <= 1e-12 * abs(_fx(b['p'])), k
            if b['scores_py2_safe']:
                for u, v in zip(a['scores'], b['scores']):
                    assert abs(u - _fx(v)) <= 1e-9 * max(1.0, abs(_fx(v))), k
            nseg += 1
        loc.close()
    assert nseg >= 6
