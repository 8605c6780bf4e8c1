"""The kernel's lane program (biseqt_amd/csrc/pw_wave.h) and host planner (pw_plan.h), executed by the
64-fiber CPU emulator of tests/emu, against the golden vectors and the oracle.  This is the CPU-side
check of the device logic: same source, lockstep lanes, emulated DPP shifts."""
import os

import numpy as np
import pytest

from tests.emu import emu
from tests.helpers import check_against_expect, dec, kw_of, load_golden


def _frame_lens(rec, kw):
    X = (kw['origin_range'][1] - kw['origin_range'][0]) if 'origin_range' in kw else len(rec['origin'])
    Y = (kw['mutant_range'][1] - kw['mutant_range'][0]) if 'mutant_range' in kw else len(rec['mutant'])
    return X, Y


def _pick_bk(rec, kw, minimum=2):
    X, Y = _frame_lens(rec, kw)
    nd = X + Y + 1 if kw['mode'] == 0 else min(kw['diag_range'][1], X) - max(kw['diag_range'][0], -Y) + 1
    for bk in (2, 4, 8, 16, 32):
        if bk >= minimum and 64 * bk >= nd:
            return bk
    return None


def test_emu_known_answers():
    for k, rec in enumerate(load_golden('known_answers.json')):
        kw = kw_of(rec)
        if len(rec['origin']) > 5000:
            continue                     # the 2e4 memory case is for the GPU
        got = emu.solve(dec(rec['origin']), dec(rec['mutant']), bk=_pick_bk(rec, kw), **kw)
        check_against_expect(got, rec['expect'], where='known[%d]' % k)


@pytest.mark.parametrize('variant', ['i32', 'f64', 'generic'])
def test_emu_random_matrix(variant):
    recs = load_golden('random_matrix.json.gz')
    step = {'i32': 7, 'f64': 13, 'generic': 11}[variant]
    ekw = {'i32': {}, 'f64': dict(use_double=True), 'generic': dict(force_generic=True)}[variant]
    n = 0
    for k in range(0, len(recs), step):
        rec = recs[k]
        kw = kw_of(rec)
        minimum = (2, 4, 8, 16, 32)[k % 5] if k % 3 == 0 else 2
        bk = _pick_bk(rec, kw, minimum)
        got = emu.solve(dec(rec['origin']), dec(rec['mutant']), bk=bk, **ekw, **kw)
        check_against_expect(got, rec['expect'], where='random[%d] %s bk=%d' % (k, variant, bk))
        n += 1
    assert n > 100


def test_emu_float_logodds():
    recs = load_golden('float_logodds.json')
    for k in range(0, len(recs), 4):
        rec = recs[k]
        kw = kw_of(rec)
        got = emu.solve(dec(rec['origin']), dec(rec['mutant']), bk=_pick_bk(rec, kw), use_double=True, **kw)
        check_against_expect(got, rec['expect'], where='float[%d]' % k)


def test_emu_steady_phase_and_score_plane(oracle):
    """Longer problems so that the unpredicated steady-phase body runs; the score-plane dump of the
    generic kernel against the oracle's table."""
    rng = np.random.default_rng(5)
    for trial in range(6):
        n = int(rng.integers(150, 400))
        o = rng.integers(0, 4, n)
        m = o.copy()
        m[rng.random(n) < 0.1] = rng.integers(0, 4, int((rng.random(n) < 0.1).sum() or 1))[0]
        m = np.delete(m, rng.integers(0, n, 5))
        kw = dict(L=4, mode=1, alntype=int(rng.integers(0, 3)), diag_range=(-int(rng.integers(6, 40)), int(rng.integers(6, 40))),
                  match=1., mismatch=-3., go=[-5., 0., -1.][trial % 3], ge=-2.)
        a = oracle.solve(o, m, **kw)
        b = emu.solve(o, m, bk=2, **kw)
        for key in ('opt', 'score', 'transcript', 'origin_idx', 'mutant_idx'):
            assert a[key] == b[key], (trial, key)
    o = rng.integers(0, 4, 40)
    m = rng.integers(0, 4, 33)
    kw = dict(L=4, mode=0, alntype=1, match=2., mismatch=-1., go=-2., ge=-1.)
    a = oracle.solve(o, m, want_table=True, **kw)
    b = emu.solve(o, m, bk=2, want_table=True, **kw)
    H = a['H'].reshape(41, 34)
    plane = b['hdump'].reshape(40 + 33 + 1, 34)
    for x in range(41):
        for y in range(34):
            assert plane[x - y + 33, min(x, y)] == H[x, y], (x, y)


def test_emu_packed16_local(oracle):
    """The packed 16-bit kernel body (WaveFill16: LOCAL / B_LOCAL only) against the oracle: steady and
    edge blocks, clamped bands, sub-word sequences, every supported BK."""
    from biseqt_amd import synth
    rng = np.random.default_rng(16)
    n = 0
    for trial in range(70):
        L = int(rng.choice([2, 4]))
        X = int(rng.integers(0, 260)) if trial % 4 else int(rng.integers(0, 12))
        o = rng.integers(0, L, X).astype(np.uint8)
        m = synth.mutate(rng, o, 0.08, 0.05, 0.3, L) if X else rng.integers(0, L, int(rng.integers(0, 9))).astype(np.uint8)
        if rng.random() < 0.3:
            m = np.concatenate([rng.integers(0, L, int(rng.integers(0, 30))).astype(np.uint8), m])
        kw = dict(L=L, match=float(rng.choice([1, 2, 5, 5, -1])), mismatch=float(rng.choice([0, -1, -3, 6])),   # also mismatch > match, match < 0
                  go=float(rng.choice([0, -1, -5])), ge=float(rng.choice([0, -1, -2])))
        if rng.random() < 0.7:
            r, c = int(rng.integers(1, 50)), int(rng.integers(-10, 10))
            kw.update(mode=1, alntype=1, diag_range=(c - r, c + r))
            nd = min(c + r, X) - max(c - r, -len(m)) + 1
        else:
            kw.update(mode=0, alntype=1)
            nd = X + len(m) + 1
        bk = next((b for b in (4, 8, 16, 32) if b >= (4, 8, 16)[trial % 3] and 64 * b >= nd), None)
        if bk is None:
            continue
        a = oracle.solve(o, m, **kw)
        b = emu.solve(o, m, bk=bk, packed16=1 + trial % 2, **kw)   # 1: lane-packed form, 2: one pair per wave
        for key in ('init_rc', 'opt', 'score', 'transcript', 'origin_idx', 'mutant_idx', 'tb_null', 'would_panick'):
            assert a.get(key) == b.get(key), (trial, key, a.get(key), b.get(key), kw, bk)
        if kw['match'] >= 0 and min(X, len(m)) * max(kw['match'], kw['mismatch'], 0) <= 2047:
            # 3: every score held times 4 (tie nibble by one three-operand add; admitted below 2048, match score >= 0)
            for pk in (3, 4):                                         # 4: the lane-packed form of it
                c = emu.solve(o, m, bk=bk, packed16=pk, **kw)
                for key in ('init_rc', 'opt', 'score', 'transcript', 'origin_idx', 'mutant_idx', 'tb_null', 'would_panick'):
                    assert a.get(key) == c.get(key), (trial, 'scaled', pk, key, a.get(key), c.get(key), kw, bk)
        n += 1
    assert n > 50


def test_emu_packed16_overlap_and_global(oracle):
    """The packed 16-bit kernel body with the overlap / global rules (WaveFill16<.., RULE = 1 | 2>: deeper sentinel,
    clamped band-top offer, begin only in a diagonal's first cell / at (0, 0), captured last cells) against the
    oracle: negative scores, clamped and infeasible bands, every supported BK, both lane layouts."""
    from biseqt_amd import synth
    rng = np.random.default_rng(17)
    n = 0
    for trial in range(120):
        L = int(rng.choice([2, 4]))
        X = int(rng.integers(0, 260)) if trial % 5 else int(rng.integers(0, 12))
        o = rng.integers(0, L, X).astype(np.uint8)
        kind = trial % 3
        if kind == 0:
            m = synth.mutate(rng, o, 0.08, 0.05, 0.3, L) if X else rng.integers(0, L, int(rng.integers(0, 9))).astype(np.uint8)
        elif kind == 1:                                     # suffix of o = prefix of m
            k = int(rng.integers(0, X + 1))
            m = np.concatenate([o[k:], rng.integers(0, L, int(rng.integers(0, 60))).astype(np.uint8)])
        else:                                               # unrelated: scores go far below zero
            m = rng.integers(0, L, int(rng.integers(0, 260))).astype(np.uint8)
        kw = dict(L=L, match=float(rng.choice([1, 2, 5])), mismatch=float(rng.choice([0, -1, -3])),
                  go=float(rng.choice([0, -1, -5])), ge=float(rng.choice([0, -1, -2])))
        alntype = 2 if trial % 2 else 0                     # B_OVERLAP / B_GLOBAL
        r, c = int(rng.integers(0, 50)), int(rng.integers(-10, 10))
        if alntype == 0 and rng.random() < 0.8:             # make the global band feasible most of the time
            lo, hi = min(0, X - len(m)) - int(rng.integers(0, 20)), max(0, X - len(m)) + int(rng.integers(0, 20))
        else:
            lo, hi = c - r, c + r
        kw.update(mode=1, alntype=alntype, diag_range=(lo, hi))
        nd = min(hi, X) - max(lo, -len(m)) + 1
        bk = next((b for b in (4, 8, 16, 32) if b >= (4, 8, 16)[trial % 3] and 64 * b >= nd), None)
        if bk is None:
            continue
        a = oracle.solve(o, m, **kw)
        b = emu.solve(o, m, bk=bk, packed16=1 + trial % 2, **kw)
        for key in ('init_rc', 'opt', 'score', 'transcript', 'origin_idx', 'mutant_idx', 'tb_null', 'would_panick'):
            assert a.get(key) == b.get(key), (trial, key, a.get(key), b.get(key), kw, bk)
        n += 1
    assert n > 90


def test_emu_packed16_standard_mode_global_and_overlap(oracle):
    """Standard-mode GLOBAL and OVERLAP are the global / overlap rules on the band [-Y, X] (OVERLAP with its own order of
    ties among the last cells): the packed rule-2 / rule-1 bodies must equal the oracle there too (empty and one-letter
    sequences, unrelated sequences whose scores go far below zero, both lane layouts)."""
    from biseqt_amd import synth
    rng = np.random.default_rng(23)
    n = 0
    for trial in range(120):
        L = int(rng.choice([2, 4]))
        X = int(rng.integers(0, 150)) if trial % 5 else int(rng.integers(0, 4))
        o = rng.integers(0, L, X).astype(np.uint8)
        if trial % 3 == 0:
            m = rng.integers(0, L, int(rng.integers(0, 150))).astype(np.uint8)
        else:
            m = synth.mutate(rng, o, 0.08, 0.05, 0.3, L) if X else rng.integers(0, L, int(rng.integers(0, 5))).astype(np.uint8)
        kw = dict(L=L, mode=0, alntype=(0, 4, 5, 6)[trial % 4], match=float(rng.choice([1, 2, 5])),    # GLOBAL / OVERLAP / START_- / END_ANCHORED_OVERLAP
                  mismatch=float(rng.choice([0, -1, -3])), go=float(rng.choice([0, -1, -5])), ge=float(rng.choice([0, -1, -2])))
        if kw['alntype'] >= 4 and trial % 3 == 1 and X > 10:      # suffix of o = prefix of m: several last cells may tie
            m = np.concatenate([o[int(rng.integers(0, X)):], rng.integers(0, L, int(rng.integers(0, 40))).astype(np.uint8)])
        nd = X + len(m) + 1
        bk = next((b for b in (4, 8, 16, 20) if 64 * b >= nd), None)
        if bk is None:
            continue
        a = oracle.solve(o, m, **kw)
        b = emu.solve(o, m, bk=bk, packed16=1 + trial % 2, **kw)
        for key in ('init_rc', 'opt', 'score', 'transcript', 'origin_idx', 'mutant_idx', 'tb_null', 'would_panick'):
            assert a.get(key) == b.get(key), (trial, key, a.get(key), b.get(key), kw, bk)
        n += 1
    assert n > 100


def test_emu_packed16_overlap_last_cell_on_a_block_boundary(oracle):
    """Regression (GPU fuzz, seed 51): a diagonal whose LAST cell is the last step of a block that the planner's steady
    range still covers -- the overlap / global rules must run the edge body there to capture it."""
    import json
    rec = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'packed_overlap_regression.json')))
    o, m = np.array(rec['origin'], np.uint8), np.array(rec['mutant'], np.uint8)
    sc = rec['scores']
    kw = dict(L=4, mode=1, alntype=rec['alntype'], diag_range=tuple(rec['band']), match=float(sc[0]), mismatch=float(sc[1]),
              go=float(sc[2]), ge=float(sc[3]))
    a = oracle.solve(o, m, **kw)
    for pk in (1, 2):
        for bk in (4, 8):
            b = emu.solve(o, m, bk=bk, packed16=pk, **kw)
            for key in ('opt', 'score', 'transcript', 'origin_idx', 'mutant_idx'):
                assert a.get(key) == b.get(key), (pk, bk, key)
    # the same geometry family: every band end so that tl_min sweeps all residues modulo 16
    for hi in range(80, 100):
        kw['diag_range'] = (-84, hi)
        a = oracle.solve(o, m, **kw)
        b = emu.solve(o, m, bk=4, packed16=2, **kw)
        assert a['opt'] == b['opt'] and a['score'] == b['score'] and a['transcript'] == b['transcript'], hi


def test_emu_strip_pipeline(oracle):
    """The strip pipeline's lane program (pw_strip.h: row strips, FIFO hand-off, end-cell reduction, ballot-driven walker)
    on the CPU emulator against the oracle: all seven standard-mode types, linear / affine gaps, tables from empty to a
    dozen strips (so that start, steady and general blocks, stale FIFO granules and both store flavours all occur)."""
    from biseqt_amd import synth
    rng = np.random.default_rng(2026)
    scores = [(1, -3, -5, -2), (1, 0, 0, 0), (2, -1, 0, -1), (5, -4, -10, -1), (1, -1, -1, -1), (1, -3, 0, -2), (1, 6, -5, -2)]
    n = 0
    for trial in range(84):
        big = trial % 6 == 0
        X = int(rng.integers(200, 800)) if big else (int(rng.integers(0, 200)) if trial % 5 else int(rng.integers(0, 6)))
        o = rng.integers(0, 4, X).astype(np.uint8)
        kind = trial % 4
        if kind == 0:
            m = synth.mutate(rng, o, 0.1, 0.05, 0.3) if X else rng.integers(0, 4, int(rng.integers(0, 9))).astype(np.uint8)
        elif kind == 1:
            k = int(rng.integers(0, X + 1))
            m = np.concatenate([o[k:], rng.integers(0, 4, int(rng.integers(0, 90))).astype(np.uint8)])
        elif kind == 2:
            m = rng.integers(0, 4, int(rng.integers(0, 300))).astype(np.uint8)
        else:
            m = o.copy()
        sc = scores[trial % len(scores)]
        kw = dict(L=4, mode=0, alntype=trial % 7, match=float(sc[0]), mismatch=float(sc[1]), go=float(sc[2]), ge=float(sc[3]))
        a = oracle.solve(o, m, **kw)
        # (both forms of the substitution score: the row's scores as 4 bytes + v_perm_b32, and compare / select)
        b = emu.solve_strip(o, m, epoch=3 + trial, byte_rows=trial % 3 != 0, **kw)
        for key in ('init_rc', 'opt', 'score', 'transcript', 'origin_idx', 'mutant_idx', 'tb_null', 'would_panick'):
            assert a.get(key) == b.get(key), (trial, key, X, len(m), kw)
        n += 1
    assert n == 84
    # a substitution matrix on the strips' byte rows (round 3)
    for trial in range(16):
        X = int(rng.integers(100, 500)); L = 4
        o = rng.integers(0, L, X).astype(np.uint8)
        m = synth.mutate(rng, o, 0.1, 0.05, 0.3) if trial % 2 else rng.integers(0, L, int(rng.integers(130, 400))).astype(np.uint8)
        S = _random_matrix(rng, L, trial % 2)
        kw = dict(L=L, mode=0, alntype=trial % 7, subst=[[float(v) for v in row] for row in S], go=float(-(trial % 3) * 3), ge=float(-1 - trial % 2))
        a = oracle.solve(o, m, **kw)
        b = emu.solve_strip(o, m, epoch=100 + trial, **kw)
        for key in ('init_rc', 'opt', 'score', 'transcript', 'origin_idx', 'mutant_idx', 'tb_null', 'would_panick'):
            assert a.get(key) == b.get(key), (trial, key, X, len(m), kw)
    # table widths around the thresholds of the block kinds (pw_strip.h, run(): a strip's first two blocks take the fast
    # hand-over when Y > 128; steady blocks are those with k0 >= 64 and k0 + 63 <= Y; the rest run as ending / general ones)
    for trial, Y in enumerate((95, 126, 127, 128, 129, 130, 158, 159, 160, 190, 191, 192, 193, 222, 223, 224, 255, 256, 287)):
        X = 150 + 7 * (trial % 5)
        o = rng.integers(0, 4, X).astype(np.uint8)
        m = synth.mutate(rng, o, 0.08, 0.04, 0.3)
        m = np.concatenate([m, rng.integers(0, 4, 300).astype(np.uint8)])[:Y]
        sc = scores[trial % len(scores)]
        kw = dict(L=4, mode=0, alntype=(1, 0, 3, 1, 2, 5)[trial % 6], match=float(sc[0]), mismatch=float(sc[1]), go=float(sc[2]), ge=float(sc[3]))
        a = oracle.solve(o, m, **kw)
        b = emu.solve_strip(o, m, epoch=200 + trial, byte_rows=trial % 2 == 0, **kw)
        for key in ('init_rc', 'opt', 'score', 'transcript', 'origin_idx', 'mutant_idx', 'tb_null', 'would_panick'):
            assert a.get(key) == b.get(key), (trial, key, X, len(m), kw)


def _random_matrix(rng, L, kind):
    """An integer substitution matrix the packed kernels admit (minimum <= 0, range <= 127; `kind` 1: range <= 31 so that
    the scores-times-4 form takes it too)."""
    hi = 7 if kind else 40
    S = rng.integers(-hi, hi + 1, size=(L, L))
    S[np.arange(L), np.arange(L)] = rng.integers(1, hi + 1, size=L)           # matches usually score
    if rng.random() < 0.3:
        S = (S + S.T) // 2                                                     # symmetric, like real matrices
    if rng.random() < 0.15:
        S[int(rng.integers(0, L)), int(rng.integers(0, L))] = hi + 2           # a mismatch that beats every match
    if S.min() > 0:
        S[0, L - 1] = 0
    return [[float(v) for v in row] for row in S]


def test_emu_packed16_substitution_matrix(oracle):
    """The packed kernel body with a substitution matrix (WaveFill16<.., MAT>: rows of bytes in the origin window, byte
    selectors in the mutant window, one byte permute per cell pair) against the oracle: rules 0 .. 3, both lane layouts,
    alphabets of 2 .. 4 letters, asymmetric matrices, clamped bands, empty and sub-word sequences."""
    from biseqt_amd import synth
    rng = np.random.default_rng(31)
    n = 0
    seen = set()
    for trial in range(160):
        L = int(rng.choice([2, 3, 4]))
        X = int(rng.integers(0, 260)) if trial % 5 else int(rng.integers(0, 12))
        o = rng.integers(0, L, X).astype(np.uint8)
        kind = trial % 3
        if kind == 0:
            m = synth.mutate(rng, o, 0.1, 0.05, 0.3, L) if X else rng.integers(0, L, int(rng.integers(0, 9))).astype(np.uint8)
        elif kind == 1 and X > 0:
            k = int(rng.integers(0, X + 1))
            m = np.concatenate([o[k:], rng.integers(0, L, int(rng.integers(0, 60))).astype(np.uint8)])
        else:
            m = rng.integers(0, L, int(rng.integers(0, 200))).astype(np.uint8)
        small = trial % 2
        subst = _random_matrix(rng, L, small)
        kw = dict(L=L, subst=subst, go=float(rng.choice([0, -1, -5])), ge=float(rng.choice([0, -1, -2])))
        which = trial % 8
        if which < 3:                                          # B_LOCAL / LOCAL
            if rng.random() < 0.7:
                r, c = int(rng.integers(1, 50)), int(rng.integers(-10, 10))
                kw.update(mode=1, alntype=1, diag_range=(c - r, c + r))
            else:
                kw.update(mode=0, alntype=1)
        elif which < 5:                                        # B_OVERLAP / B_GLOBAL
            alntype = 2 if which == 3 else 0
            if alntype == 0 and rng.random() < 0.8:
                lo, hi = min(0, X - len(m)) - int(rng.integers(0, 20)), max(0, X - len(m)) + int(rng.integers(0, 20))
            else:
                r, c = int(rng.integers(0, 50)), int(rng.integers(-10, 10))
                lo, hi = c - r, c + r
            kw.update(mode=1, alntype=alntype, diag_range=(lo, hi))
        else:                                                  # standard GLOBAL / OVERLAP / START_- / END_ANCHORED_OVERLAP
            kw.update(mode=0, alntype=(0, 4, 5, 6)[trial % 4])
            if X + len(m) + 1 > 64 * 20:
                continue
        if kw['mode'] == 1:
            nd = min(kw['diag_range'][1], X) - max(kw['diag_range'][0], -len(m)) + 1
        else:
            nd = X + len(m) + 1
        bk = next((b for b in (4, 8, 16, 20, 32) if b >= (4, 8, 16)[trial % 3] and 64 * b >= nd), None)
        if bk is None:
            continue
        a = oracle.solve(o, m, **kw)
        modes = [1 + (trial // 8) % 2]
        if which < 3 and small and min(X, len(m)) * max(max(map(max, subst)), 0) <= 2047:
            modes += [3, 4]                                    # the scores-times-4 form, both layouts
        for pk in modes:
            b = emu.solve(o, m, bk=bk, packed16=pk, **kw)
            for key in ('init_rc', 'opt', 'score', 'transcript', 'origin_idx', 'mutant_idx', 'tb_null', 'would_panick'):
                assert a.get(key) == b.get(key), (trial, pk, key, a.get(key), b.get(key), kw, bk)
            seen.add((which, pk))
        n += 1
    assert n > 120 and len(seen) >= 16, (n, sorted(seen))


def test_emu_packed16_matrix_equals_match_mismatch_form(oracle):
    """Match / mismatch scores written out as a 4 x 4 matrix whose fourth letter (which never occurs in the sequences) has
    other scores: the batch is a matrix batch -- it takes the matrix form -- and must give the plain form's results."""
    from biseqt_amd import synth
    rng = np.random.default_rng(32)
    for trial in range(24):
        o = rng.integers(0, 3, int(rng.integers(30, 300))).astype(np.uint8)
        m = synth.mutate(rng, o, 0.1, 0.05, 0.3, 3)
        match, mismatch = float(rng.choice([1, 2, 5])), float(rng.choice([0, -1, -3]))
        subst = [[match if i == j else mismatch for j in range(4)] for i in range(4)]
        subst[3][0] = subst[0][3] = mismatch - 2.0
        r, c = int(rng.integers(1, 40)), int(rng.integers(-8, 8))
        base = dict(L=4, mode=1, alntype=trial % 3, diag_range=(min(c - r, 0, len(o) - len(m)), max(c + r, 0, len(o) - len(m))),
                    go=float(rng.choice([0, -2, -5])), ge=float(rng.choice([-1, -2])))
        a = emu.solve(o, m, bk=8, packed16=2, subst=subst, **base)
        b = emu.solve(o, m, bk=8, packed16=2, match=match, mismatch=mismatch, **base)
        ref = oracle.solve(o, m, match=match, mismatch=mismatch, **base)
        for key in ('opt', 'score', 'transcript', 'origin_idx', 'mutant_idx'):
            assert a.get(key) == ref.get(key) == b.get(key), (trial, key)


def test_emu_packed16_anchored_rules(oracle):
    """START_ANCHORED (begin at (0, 0), end at the first best cell, which must beat 0) and END_ANCHORED (begin anywhere,
    end at (X, Y)) on their own packed instantiations (WaveFill16 rules 5 and 4) against the oracle: related, unrelated
    and suffix-prefix pairs, empty sequences, negative match scores, both lane layouts."""
    from biseqt_amd import synth
    rng = np.random.default_rng(45)
    n = 0
    for trial in range(160):
        L = int(rng.choice([2, 4]))
        X = int(rng.integers(0, 200)) if trial % 5 else int(rng.integers(0, 4))
        o = rng.integers(0, L, X).astype(np.uint8)
        kind = trial % 4
        if kind == 0:
            m = rng.integers(0, L, int(rng.integers(0, 200))).astype(np.uint8)
        elif kind == 1 and X > 10:
            k = int(rng.integers(0, X))
            m = np.concatenate([o[k:], rng.integers(0, L, int(rng.integers(0, 60))).astype(np.uint8)])
        elif kind == 2 and X > 10:
            k = int(rng.integers(1, X))
            m = np.concatenate([rng.integers(0, L, int(rng.integers(0, 40))).astype(np.uint8), o[:k]])
        else:
            m = synth.mutate(rng, o, 0.08, 0.05, 0.3, L) if X else rng.integers(0, L, int(rng.integers(0, 5))).astype(np.uint8)
        kw = dict(L=L, mode=0, alntype=2 + trial % 2, match=float(rng.choice([1, 2, 5, -1])), mismatch=float(rng.choice([0, -1, -3, 2])),
                  go=float(rng.choice([0, -1, -5])), ge=float(rng.choice([0, -1, -2])))
        nd = X + len(m) + 1
        bk = next((b for b in (4, 8, 12, 16, 20) if b >= (4, 8, 12)[trial % 3] and 64 * b >= nd), None)
        if bk is None:
            continue
        a = oracle.solve(o, m, **kw)
        b = emu.solve(o, m, bk=bk, packed16=1 + (trial // 2) % 2, **kw)
        for key in ('init_rc', 'opt', 'score', 'transcript', 'origin_idx', 'mutant_idx', 'tb_null', 'would_panick'):
            assert a.get(key) == b.get(key), (trial, key, a.get(key), b.get(key), kw, bk, X, len(m))
        n += 1
    assert n > 140


def test_emu_multi_wavefront_kernels(oracle):
    """The multi-wavefront kernels (pw_device.h: k_fill16_mw, k_fill_mw -- a workgroup of 2 .. 8 wavefronts as one long row of
    lanes, for bands wider than a wavefront holds and for the low-latency layout of a few pairs) had no CPU coverage until
    round 3: the emulator now runs 64 x wavefronts fibers, the cross-wavefront hand-over being a neighbour exchange.  Packed
    16-bit body (plain, scores times 4, matrix form, overlap / global / anchored rules), 32-bit and f64 bodies, on banded
    and standard-mode problems whose diagonals do not fit one wavefront at the chosen lane width -- among them the shape of the
    round's fuzz find (origin much shorter than mutant, band far below the main diagonal), with admissible scores."""
    from biseqt_amd import synth
    rng = np.random.default_rng(20261009)
    n = 0
    for trial in range(44):
        kind = trial % 4
        if kind == 0:                                           # related pair, wide band
            X = int(rng.integers(150, 420)); o = rng.integers(0, 4, X).astype(np.uint8)
            m = synth.mutate(rng, o, 0.08, 0.04, 0.3)
            dr = (-int(rng.integers(100, len(m))), int(rng.integers(80, X)))
        elif kind == 1:                                         # origin much shorter than mutant, band far below the main diagonal
            X = int(rng.integers(40, 90)); o = rng.integers(0, 4, X).astype(np.uint8)
            m = rng.integers(0, 4, int(rng.integers(500, 700))).astype(np.uint8)
            k = int(rng.integers(200, 400)); m[k:k + X] = o
            dr = (-len(m) + int(rng.integers(0, 60)), int(rng.integers(0, X)))
        elif kind == 2:                                         # mutant much shorter
            m = rng.integers(0, 4, int(rng.integers(40, 90))).astype(np.uint8)
            o = rng.integers(0, 4, int(rng.integers(450, 650))).astype(np.uint8)
            dr = (-int(rng.integers(0, len(m))), len(o) - int(rng.integers(0, 60)))
        else:                                                   # standard mode: every diagonal
            X = int(rng.integers(130, 300)); o = rng.integers(0, 4, X).astype(np.uint8)
            m = synth.mutate(rng, o, 0.1, 0.05, 0.3)
            dr = None
        mode = 0 if dr is None else 1
        alntype = [1, 0, 2, 3, 4, 5, 6][trial % 7] if mode == 0 else [1, 2, 0][trial % 3]
        if mode == 1 and alntype == 0:                          # B_GLOBAL: the band must hold both corners
            dr = (min(dr[0], len(o) - len(m), 0), max(dr[1], len(o) - len(m), 0))
        ndiag = (len(o) + len(m) + 1) if dr is None else (min(dr[1], len(o)) - max(dr[0], -len(m)) + 1)
        bk = 4 if ndiag <= 2048 else 8
        waves = (ndiag + 64 * bk - 1) // (64 * bk)
        if waves < 2:
            bk, waves = 4, 2
        body = trial % 5                                        # 0 / 1 packed, 2 packed matrix, 3 32-bit, 4 f64
        kw = dict(L=4, mode=mode, alntype=alntype, go=float(-(trial % 3) * 2), ge=float(-1 - trial % 2))
        if dr is not None:
            kw['diag_range'] = dr
        if body == 2:
            kw['subst'] = _random_matrix(rng, 4, 1)
        else:
            sc = [(1, -3), (2, -1), (1, 0), (5, -4)][trial % 4]
            kw.update(match=float(sc[0]), mismatch=float(sc[1]))
        ek = dict(bk=bk, waves=waves)
        if body <= 2:
            ek['packed16'] = 3 if (body == 1 and mode == 1 and alntype == 1) or (body == 1 and mode == 0 and alntype == 1) else 2
        elif body == 4:
            ek['use_double'] = True
        a = oracle.solve(o, m, **kw)
        b = emu.solve(o, m, **kw, **ek)
        for key in ('init_rc', 'opt', 'score', 'transcript', 'origin_idx', 'mutant_idx', 'tb_null', 'would_panick'):
            assert a.get(key) == b.get(key), (trial, key, len(o), len(m), kw, ek)
        n += 1
    assert n == 44
