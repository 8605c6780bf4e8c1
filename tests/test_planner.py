"""The host planner without a GPU (`pw_plan_only`: planning is pure host arithmetic, split from device allocation): the
kernel choices DESIGN.md describes are pinned here shape by shape, so that a change of the timing model
(biseqt_amd/csrc/pw_model.h) or of an admission rule shows up as a failing line, not as a silent slowdown.  Whether each
choice is still the FASTEST one is the GPU's to say: tests/test_gpu_planner.py."""
import os
import pytest

from biseqt_amd import _pwlib as W
from biseqt_amd.batch import plan_only

CFG = dict(match_score=1, mismatch_score=-3, go_score=-5, ge_score=-2)
BLASTISH = [[1, -3, -2, -3], [-3, 1, -3, -2], [-2, -3, 1, -3], [-3, -2, -3, 1]]


def test_baseline_configs():
    # config 1: one 1 kb x 1 kb standard-mode pair -- the strips as a batch call, a workgroup with the score plane for the
    # drop-in calls (table_scores needs every cell's score)
    r = plan_only([(1000, 1000)], alnmode=0, alntype=0, **CFG)
    assert r['kernel'].startswith('k_fill_strip') and r['strips'] == 1
    r = plan_only([(1000, 1000)], alnmode=0, alntype=0, flags=W.PW_FLAG_DUMP_SCORES, **CFG)
    assert r['kernel'].startswith('k_fill_mw<int, 4') and r['workgroup'] == 1
    # config 2: the packed kernel, scores held times 4, one pair per wavefront, 8 diagonals per lane
    r = plan_only([(2000, 2010, -200, 200)] * 10000, alnmode=1, alntype=1, **CFG)
    assert r['kernel'] == 'k_fill16<8, false> x4 matrix' and r['one_wavefront'] == 10000 and r['score_dtype'] == 'i32'
    # (match / mismatch over 4 letters runs on the matrix form where that measured faster; the knob restores the plain form)
    os.environ['PWLIB_SIMPLE_AS_MATRIX'] = '0'
    try:
        r = plan_only([(2000, 2010, -200, 200)] * 10000, alnmode=1, alntype=1, **CFG)
    finally:
        os.environ.pop('PWLIB_SIMPLE_AS_MATRIX')
    assert r['kernel'] == 'k_fill16<8, false> x4' and not r['matrix']
    # ... not lane-packed, not under the overlap rule
    assert plan_only([(2000, 2010, -200, 200)] * 10000, alnmode=1, alntype=2, **CFG)['kernel'] == 'k_fill16<8, false, 1>'
    assert plan_only([(300, 300, -10, 10)] * 20000, alnmode=1, alntype=1, **CFG)['kernel'] == 'k_fill16<4, true> x4'
    assert plan_only([(2000, 2010, -400, 400)] * 3000, alnmode=1, alntype=1, **CFG)['kernel'] == 'k_fill16<16, false> x4 matrix'
    assert plan_only([(1000, 1000)] * 5000, alnmode=0, alntype=0, **CFG)['kernel'] == 'k_fill16<32, false, 2> matrix'
    # config 3: the strip pipeline
    r = plan_only([(100000, 100218)], alnmode=0, alntype=1, **CFG)
    assert r['kernel'] == 'k_fill_strip<true> x row strips'
    # config 4's alignments: the packed overlap rule
    r = plan_only([(5000, 5000, 2700, 3300)] * 200000, alnmode=1, alntype=2, **CFG)
    assert r['kernel'].startswith('k_fill16<') and r['packed_rule'] == 1
    # config 5's extensions: 1 / p_min - 1 = 0.25 is dyadic -> integer kernels; 3/7 is not -> f64
    r = plan_only([(12000, 12100, -300, 300)] * 50, alnmode=1, alntype=0, match_score=0.25, mismatch_score=-1, go_score=0, ge_score=-1)
    assert r['score_dtype'] == 'i32' and r['scale_shift'] == 2
    r = plan_only([(12000, 12100, -300, 300)] * 50, alnmode=1, alntype=0, match_score=1 / .7 - 1, mismatch_score=-1, go_score=0, ge_score=-1)
    assert r['score_dtype'] == 'f64' and r['scale_shift'] == 0


def test_few_pairs_one_after_another_or_all_at_once():
    """The timing model's decision (pw_model.h): one 2 kb pair 0.6 ms on the strips against 1.5 ms on the packed workgroups;
    four pairs 2.4 ms against 1.6 ms (profiles/round2_e_few_pairs.txt)."""
    kw = dict(alnmode=0, alntype=1, **CFG)
    assert plan_only([(2000, 2000)], **kw)['strips'] == 1
    r = plan_only([(2000, 2000)] * 4, **kw)
    assert r['kernel'].startswith('k_fill16_mw<8, 3>') and r['workgroup'] == 4
    assert plan_only([(8000, 8000)], **kw)['strips'] == 1
    assert plan_only([(2000, 2000)] * 1000, **kw)['kernel'].startswith('k_fill16_mw<8, 3>')
    # f64 scores: the strips do not serve them -- one wide pair goes to the tiled kernel, many to narrow-lane workgroups
    f = dict(alnmode=0, alntype=1, match_score=0.3, mismatch_score=-1.1, go_score=-2, ge_score=-0.7)
    assert plan_only([(8000, 8000)], **f)['tiled'] == 1
    r = plan_only([(3000, 3000, -600, 600)] * 3000, alnmode=1, alntype=1, match_score=0.3, mismatch_score=-1.1, go_score=-2, ge_score=-0.7)
    assert r['kernel'].startswith('k_fill_mw<double, 4') and '5 wavefronts' in r['kernel']
    # PW_FLAG_NO_STRIP (what a repaired strip pair is solved with) keeps wide pairs off the strips
    assert plan_only([(100000, 100218)], flags=W.PW_FLAG_NO_STRIP, **kw)['strips'] == 0
    assert plan_only([(2000, 2000)], flags=W.PW_FLAG_NO_STRIP, **kw)['strips'] == 0


def test_lane_layouts():
    kw = dict(alnmode=1, alntype=1, **CFG)
    # narrow bands: several pairs per wavefront, the fewest diagonals per lane
    # (2000 pairs: between "fewer pairs than SIMDs" and "enough wavefronts after packing" -- the gap planner_check.py found)
    for shapes in ([(100, 100, -10, 10)] * 20000, [(1000, 1000, -10, 10)] * 1000, [(2000, 2000, -20, 20)] * 16,
                   [(1000, 1000, -10, 10)] * 2000):
        assert plan_only(shapes, **kw)['kernel'] == 'k_fill16<4, true> x4', shapes[0]
    # bands wider than one wavefront holds, many pairs: workgroups with narrow lanes (32 diagonals per lane spill)
    r = plan_only([(10000, 10000, -1500, 1500)] * 300, **kw)
    assert r['kernel'].startswith('k_fill_mw<int, 8') and '6 wavefronts' in r['kernel']
    # a band of one diagonal over a million letters (the reference's memory test, tests/test_pw.py:95-103)
    r = plan_only([(1000000, 1000000, 0, 0)], alnmode=1, alntype=0, match_score=1, mismatch_score=0)
    assert r['kernel'] == 'k_fill<int, 2, false, false, false>'


@pytest.mark.parametrize('alntype,rule', [(0, 2), (1, 0), (2, 5), (3, 4), (4, 1), (5, 1), (6, 1)])
def test_every_standard_type_has_a_packed_kernel(alntype, rule):
    r = plan_only([(500, 510)] * 1000, alnmode=0, alntype=alntype, match_score=2, mismatch_score=-3, go_score=-4, ge_score=-1)
    assert r['kernel'].startswith('k_fill16<'), r
    assert r['packed_rule'] in ((rule, 3) if rule == 0 else (rule,)), r


def test_scoring_surface_admissions():
    base = dict(alnmode=1, alntype=1)
    shapes = [(2000, 2010, -200, 200)] * 1000
    r = plan_only(shapes, subst_scores=BLASTISH, go_score=-5, ge_score=-2, **base)
    assert r['kernel'] == 'k_fill16<8, false> x4 matrix' and r['matrix']
    # a matrix the packed kernels do not admit (five letters; BLOSUM-like, 20 letters): the fast 32-bit kernel -- every
    # wavefront kernel reads its substitution scores from a table in LDS -- not the generic one; 40 letters (beyond the LDS
    # copy) and go > 0: the generic kernel; scores too large for 16 bits: the 32-bit kernel; not dyadic: f64
    for L in (5, 20):
        rL = plan_only(shapes, alphabet_len=L, subst_scores=[[4 if i == j else -1 - (i + j) % 3 for j in range(L)] for i in range(L)],
                       go_score=-5, ge_score=-2, **base)
        assert rL['kernel'] == 'k_fill<int, 8, true, true, false>', (L, rL)
    r40 = plan_only(shapes, alphabet_len=40, subst_scores=[[4 if i == j else -1 - (i + j) % 3 for j in range(40)] for i in range(40)],
                    go_score=-5, ge_score=-2, **base)
    assert r40['kernel'] == 'k_fill<int, 8, false, true, true>'
    logodds = [[0.8 if i == j else -1.7 + 0.01 * (i + j) for j in range(4)] for i in range(4)]
    assert plan_only(shapes, subst_scores=logodds, go_score=-0.69, ge_score=-1.2, **base)['kernel'] == 'k_fill<double, 8, true, true, false>'
    assert plan_only(shapes, match_score=1, mismatch_score=-3, go_score=2, ge_score=-2, **base)['kernel'] == 'k_fill<int, 8, false, true, true>'
    assert plan_only(shapes, match_score=100, mismatch_score=-300, go_score=-500, ge_score=-200, **base)['kernel'] == 'k_fill<int, 8, true, true, false>'
    assert plan_only(shapes, match_score=0.1, mismatch_score=-1, go_score=0, ge_score=-1, **base)['score_dtype'] == 'f64'
    # forcing flags
    assert plan_only(shapes, flags=W.PW_FLAG_NO_PACKED16, **base, **CFG)['kernel'] == 'k_fill<int, 8, true, true, false>'
    assert plan_only(shapes, flags=W.PW_FLAG_FORCE_F64, **base, **CFG)['kernel'] == 'k_fill<double, 8, true, true, false>'
    assert plan_only(shapes, flags=W.PW_FLAG_FORCE_GENERIC, **base, **CFG)['kernel'] == 'k_fill<int, 8, false, true, true>'


def test_planning_errors_are_reported():
    with pytest.raises(RuntimeError, match='negative'):
        plan_only([(-1, 5)], alnmode=0, alntype=0)
    with pytest.raises(RuntimeError, match='type'):
        plan_only([(5, 5)], alnmode=0, alntype=9)


def test_positive_mismatch_scores_never_take_the_plain_packed_form():
    """The plain form of the packed kernels scores letters outside a sequence as a mismatch; with a mismatch score above 0 (the
    API accepts it) cells that wait for their diagonal's first cell would creep up from the 16-bit sentinel.  The matrix form
    (off-table letters score the matrix minimum, <= 0) takes such scores where it applies, the 32-bit kernels otherwise."""
    shapes = {'config 2': ([(2000, 2010, -200, 200)] * 10000, dict(alnmode=1, alntype=1)),
              'overlap': ([(2000, 2010, -200, 200)] * 10000, dict(alnmode=1, alntype=2)),
              'lane-packed': ([(300, 300, -10, 10)] * 20000, dict(alnmode=1, alntype=1)),
              'wide band': ([(524, 3656, -3414, 258)] * 3, dict(alnmode=1, alntype=1)),
              'standard': ([(1000, 1000)] * 5000, dict(alnmode=0, alntype=0))}
    for tag, (sh, kw) in shapes.items():
        for L in (1, 4, 20):
            for sc in ((1, 6, -5, -2), (2, 3, -1, -1), (1, 100, -5, -2)):
                k = plan_only(sh, alphabet_len=L, match_score=sc[0], mismatch_score=sc[1], go_score=sc[2], ge_score=sc[3], **kw)['kernel']
                assert 'k_fill16' not in k, (tag, L, sc, k)
            k = plan_only(sh, alphabet_len=L, match_score=-1, mismatch_score=2, go_score=-2, ge_score=-1, **kw)['kernel']
            # (with ONE letter the mismatch score never occurs and off-table letters score the match score, here -1)
            assert 'k_fill16' not in k or (L == 4 and 'matrix' in k) or L == 1, (tag, L, k)
            # ... and a mismatch score of 0 or below keeps the packed kernels
            k = plan_only(sh, alphabet_len=max(L, 2), match_score=1, mismatch_score=0, go_score=-2, ge_score=-1, **kw)['kernel']
            assert 'k_fill16' in k or 'strip' in k, (tag, L, k)


def test_lane_layout_is_priced_not_just_packed():
    """Many pairs with narrow bands of MIXED widths (config 4's alignment stage: 9 .. 111 diagonals, mean 77): the layout that
    keeps most slots busy (28 diagonals per lane, 16 pairs per wavefront) leaves 1250 wavefronts for 1024 SIMDs at 20 000 pairs
    and measured 7.4 ms against 4.9 ms at 8 per lane; with 50 000 pairs 16 per lane win (profiles/round3_n_lane_width.txt,
    round3_o_layout_race.txt).  pw_model.h prices slot cost, busy share and the last round of wavefronts."""
    import numpy as np
    rng = np.random.default_rng(5)
    kw = dict(alnmode=1, alntype=2, alphabet_len=4, **CFG)

    def shapes(n):
        r = rng.integers(4, 56, n)
        r[0] = 55
        return [(5000, 5000, -int(v), int(v)) for v in r]
    assert plan_only(shapes(20000), **kw)['kernel'] == 'k_fill16<8, true, 1>'
    assert plan_only(shapes(50000), **kw)['kernel'] == 'k_fill16<16, true, 1>'
    # one band width, 81 diagonals: the narrowest lanes (three pairs per wavefront)
    assert plan_only([(300, 300, -40, 40)] * 20000, alnmode=1, alntype=1, alphabet_len=4, **CFG)['kernel'] == 'k_fill16<4, true> x4'
