"""The C-ABI boundary without a GPU: the shared object loads, exports every symbol the headers
declare, the struct layouts match the reference's, and dptable_init (pure host arithmetic) agrees
with the reference on dimensions, band clamp and feasibility."""
import ctypes as C
import os
import re

import pytest

from biseqt_amd import _pwlib as W
from tests.helpers import dec, kw_of, load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions(header):
    txt = open(os.path.join(ROOT, 'include', header)).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    txt = re.sub(r'//[^\n]*', '', txt)
    return set(re.findall(r'\b(dptable_\w+|pw_\w+)\s*\(', txt)) - {'pw_batch', 'pw_seed_index', 'pw_read_pair', 'pw_overlap_band'}


def test_library_loads_and_exports_every_declared_symbol():
    lib = W.load()
    declared = _declared_functions('pwlib.h') | _declared_functions('pw_batch.h')
    assert {'dptable_init', 'dptable_solve', 'dptable_traceback', 'dptable_free'} <= declared
    assert declared == set(W.EXPORTS), declared ^ set(W.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    seeds = _declared_functions('pw_seeds.h')
    assert seeds == set(W.SEED_EXPORTS), seeds ^ set(W.SEED_EXPORTS)
    for name in seeds:
        assert hasattr(lib, name), name
    ov = _declared_functions('pw_overlap.h')
    assert ov == set(W.OVERLAP_EXPORTS), ov ^ set(W.OVERLAP_EXPORTS)
    for name in ov:
        assert hasattr(lib, name), name
    assert C.sizeof(W.pw_read_pair) == 24


def test_struct_layouts_match_reference_abi():
    W.check_layout()     # sizeof list measured on the reference
    assert W.alnchoice.score.offset == 8 and W.alnchoice.base.offset == 16
    assert W.alnprob.mode.offset == 20 and W.alnprob.params.offset == 24
    assert W.dptable.row_lens.offset == 16 and W.dptable.prob.offset == 24
    assert W.alignment.transcript.offset == 16


def test_header_is_cdef_clean():
    """pw.py:61-66 feeds the header to cffi after dropping only lines that START with '#define'."""
    lines = open(os.path.join(ROOT, 'include', 'pwlib.h')).read().split('\n')
    kept = [l for l in lines if not l.startswith('#define')]
    assert not any(l.lstrip().startswith('#') for l in kept)


def test_enum_values_match_reference():
    assert (W.STD_MODE, W.BANDED_MODE) == (0, 1)
    assert [W.GLOBAL, W.LOCAL, W.START_ANCHORED, W.END_ANCHORED, W.OVERLAP,
            W.START_ANCHORED_OVERLAP, W.END_ANCHORED_OVERLAP] == list(range(7))
    assert [W.B_GLOBAL, W.B_LOCAL, W.B_OVERLAP] == [0, 1, 2]


def test_dptable_init_matches_reference(capfd):
    """init rc, clamped band (written back into the caller's struct), num_rows and row_lens for the
    golden problems -- dptable_init does no GPU work."""
    from oracle import ref_driver as R
    lib = R.load(W.PWLIB_SO)           # the same ctypes driver that drives the compiled reference
    recs = load_golden('random_matrix.json.gz')
    n = 0
    for k in range(0, len(recs), 3):
        rec = recs[k]
        kw = kw_of(rec)
        P = R.Problem(dec(rec['origin']), dec(rec['mutant']), **kw)
        rc = lib.dptable_init(C.byref(P.table))
        exp = rec['expect']
        assert rc == exp['init_rc'], (k, rc, exp)
        if kw['mode'] == 1:
            assert [P.params.dmin, P.params.dmax] == exp['band'], k
        if rc == 0:
            assert P.table.num_rows == exp['num_rows'], k
            X = P.frame.origin_range.j - P.frame.origin_range.i
            Y = P.frame.mutant_range.j - P.frame.mutant_range.i
            for i in range(P.table.num_rows):
                d = P.params.dmin + i if kw['mode'] == 1 else 0
                want = Y + 1 if kw['mode'] == 0 else 1 + min(d, 0) + min(X - d, Y)
                assert P.table.row_lens[i] == want
                if P.table.row_lens[i] > 0:
                    assert P.table.cells[i][0].num_choices == 0       # all cells start empty
            lib.dptable_free(C.byref(P.table))
            assert not P.table.cells
            n += 1
    assert n > 300
    C.CDLL(None).fflush(None)
    capfd.readouterr()


def test_unsupported_problems_fail_loudly(capfd):
    from oracle import ref_driver as R
    lib = R.load(W.PWLIB_SO)
    P = R.Problem([0, 1, 2], [0, 1], L=4, max_new_mins=3)
    assert lib.dptable_init(C.byref(P.table)) == -1
    # a table wider than the widest kernel (2^21 diagonals) is refused, not handed to a CPU path
    P = R.Problem([0], [0], L=4)
    P.frame.origin_range.j = 1 << 21       # only the frame lengths are read by dptable_init
    P.frame.mutant_range.j = 1 << 20
    assert lib.dptable_init(C.byref(P.table)) == -1
    # a large standard-mode table is accepted without allocating its cells on the host (lazy rows)
    P = R.Problem([0], [0], L=4)
    P.frame.origin_range.j = 100000
    P.frame.mutant_range.j = 100000
    assert lib.dptable_init(C.byref(P.table)) == 0
    assert P.table.num_rows == 100001 and P.table.row_lens[5] == 100001 and not P.table.cells[5]
    lib.dptable_free(C.byref(P.table))
    C.CDLL(None).fflush(None)
    err = capfd.readouterr().err
    assert 'max_new_mins' in err and 'no CPU fallback' in err


def test_product_does_not_import_the_oracle():
    """The product path must never route through oracle/ (or the emulator)."""
    pkg = os.path.join(ROOT, 'biseqt_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.cpp', '.h', '.hip')):
                txt = open(os.path.join(dirpath, f)).read()
                assert 'import oracle' not in txt and 'from oracle' not in txt, f
                assert 'pw_oracle' not in txt and 'emu_wave' not in txt, f


def test_seed_and_overlap_entry_points_reject_bad_input_loudly():
    """Argument validation happens before any device work: NULL / -1 plus a message, never a silent fallback."""
    import numpy as np
    lib = W.load()
    s = np.array([0, 1, 2, 3, 0, 1], np.uint8)
    bad = np.array([0, 1, 9, 3], np.uint8)
    none = C.POINTER(C.c_uint64)()
    assert not lib.pw_seeds_create(0, s.ctypes.data, len(s), s.ctypes.data, len(s), 0, 3, none, 0, -1)
    assert b'alphabet_len' in lib.pw_seeds_last_error()
    assert not lib.pw_seeds_create(0, s.ctypes.data, len(s), s.ctypes.data, len(s), 4, 40, none, 0, -1)
    assert b'wordlen' in lib.pw_seeds_last_error()
    assert not lib.pw_seeds_create(0, s.ctypes.data, len(s), s.ctypes.data, len(s), 4, 31, none, 0, -1)
    assert b'2^62' in lib.pw_seeds_last_error()
    assert not lib.pw_seeds_create(0, bad.ctypes.data, len(bad), s.ctypes.data, len(s), 4, 3, none, 0, 0)
    assert b'letter outside the alphabet' in lib.pw_seeds_last_error()
    assert lib.pw_seeds_build(None, 0, None) == -1
    rp = (W.pw_read_pair * 1)(W.pw_read_pair(0, 4, 4, 10))          # the second read runs past the arena
    out = np.zeros(64, np.uint8)
    assert lib.pw_overlap_bands(0, s.ctypes.data, len(s), rp, 1, 4, 3, 1.1, 0.7, 1. / 64, out.ctypes.data) == -1
    assert b'outside the arena' in lib.pw_overlap_last_error()
    assert lib.pw_overlap_bands(0, s.ctypes.data, len(s), rp, 1, 4, 3, 0.0, 0.7, 1. / 64, out.ctypes.data) == -1
    assert b'coefficients' in lib.pw_overlap_last_error()
