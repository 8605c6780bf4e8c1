"""ctypes wrapper of the CPU lane emulator (tests/emu/emu_wave.cpp) -- TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, 'libemu_wave.so')
SRC = os.path.join(HERE, 'emu_wave.cpp')
CSRC = os.path.join(os.path.dirname(os.path.dirname(HERE)), 'biseqt_amd', 'csrc')

_lib = None


def build(force=False):
    deps = [SRC] + [os.path.join(CSRC, f) for f in ('pw_wave.h', 'pw_strip.h', 'pw_plan.h', 'pw_types.h')]
    if force or not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in deps):
        subprocess.check_call(['g++', '-O1', '-std=c++17', '-shared', '-fPIC', '-ffp-contract=off',
                               SRC, '-o', SO])


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(SO)
        _lib.emu_solve.restype = C.c_int
        _lib.emu_solve_strip.restype = C.c_int
    return _lib


def solve(origin, mutant, mode=0, alntype=0, subst=None, L=None, match=1., mismatch=0., go=0., ge=0.,
          diag_range=None, origin_range=None, mutant_range=None, use_double=False, force_generic=False,
          bk=8, want_table=False, packed16=False, waves=1, **_):
    o = np.asarray(origin, dtype=np.int32)
    m = np.asarray(mutant, dtype=np.int32)
    if L is None:
        L = len(subst) if subst is not None else int(max([0] + list(o) + list(m))) + 1
    if subst is None:
        subst = [[match if i == j else mismatch for i in range(L)] for j in range(L)]
    S = np.ascontiguousarray(np.asarray(subst, dtype=np.float64).reshape(L, L))
    orange = origin_range if origin_range is not None else (0, len(o))
    mrange = mutant_range if mutant_range is not None else (0, len(m))
    of = np.ascontiguousarray(o[orange[0]:orange[1]])
    mf = np.ascontiguousarray(m[mrange[0]:mrange[1]])
    X, Y = len(of), len(mf)
    if X == 0:
        of = np.zeros(1, np.int32)
    if Y == 0:
        mf = np.zeros(1, np.int32)
    dr = diag_range if diag_range is not None else (0, 0)
    info = (C.c_int * 10)()
    score = C.c_double(0)
    txcap = X + Y + 2
    txbuf = C.create_string_buffer(txcap)
    hd = None
    hp = None
    if want_table:
        nd = X + Y + 1
        hd = np.zeros(nd * (min(X, Y) + 1), np.float64)
        hp = hd.ctypes.data_as(C.POINTER(C.c_double))
    lib().emu_set_waves(int(waves))
    rc = lib().emu_solve(mode, alntype, of.ctypes.data_as(C.POINTER(C.c_int)), X,
                         mf.ctypes.data_as(C.POINTER(C.c_int)), Y, L,
                         S.ctypes.data_as(C.POINTER(C.c_double)), C.c_double(go), C.c_double(ge),
                         int(dr[0]), int(dr[1]), int(use_double), int(force_generic), bk,
                         info, C.byref(score), txbuf, txcap, hp, int(packed16))
    if rc != 0:
        raise ValueError('emu_solve rc=%d' % rc)
    out = dict(init_rc=info[0], opt=None, score=None, transcript=None, origin_idx=None,
               mutant_idx=None, tb_null=None, would_panick=None)
    if mode == 1:
        out['band'] = (info[1], info[2])
    if info[0] != 0:
        return out
    out['num_rows'] = info[3]
    ex, ey = info[4], info[5]
    if ex < 0:
        out['opt'] = (-1, -1)
        return out
    out['opt'] = (ex, ey)
    out['score'] = score.value
    st = info[9]
    out['would_panick'] = bool(st & 4)
    out['tb_null'] = bool(st & 2) and not (st & 4)
    if not (st & 4) and not (st & 2):
        out['transcript'] = txbuf.value.decode('ascii')
        out['origin_idx'] = info[6] + orange[0]
        out['mutant_idx'] = info[7] + mrange[0]
    if want_table:
        out['hdump'] = hd
    return out


def solve_strip(origin, mutant, alntype=0, match=1., mismatch=0., go=0., ge=0., epoch=7, byte_rows=True, subst=None, **_):
    """Standard-mode problem through the strip pipeline (pw_strip.h): fill strip by strip, end-cell reduction, strip
    walker, fix-up.  Same result dict as :func:`solve`."""
    of = np.ascontiguousarray(np.asarray(origin, dtype=np.int32))
    mf = np.ascontiguousarray(np.asarray(mutant, dtype=np.int32))
    X, Y = len(of), len(mf)
    if X == 0:
        of = np.zeros(1, np.int32)
    if Y == 0:
        mf = np.zeros(1, np.int32)
    info = (C.c_int * 10)()
    score = C.c_double(0)
    txcap = X + Y + 2
    txbuf = C.create_string_buffer(txcap)
    lib().emu_set_strip_byte_rows(1 if byte_rows else 0)
    if subst is not None:
        S = np.ascontiguousarray(np.asarray(subst, dtype=np.float64))
        lib().emu_set_strip_matrix(S.ctypes.data_as(C.POINTER(C.c_double)), int(S.shape[0]))
    else:
        lib().emu_set_strip_matrix(None, 0)
    rc = lib().emu_solve_strip(alntype, of.ctypes.data_as(C.POINTER(C.c_int)), X, mf.ctypes.data_as(C.POINTER(C.c_int)), Y,
                               C.c_double(match), C.c_double(mismatch), C.c_double(go), C.c_double(ge), C.c_uint(epoch),
                               info, C.byref(score), txbuf, txcap)
    if rc != 0:
        raise ValueError('emu_solve_strip rc=%d' % rc)
    out = dict(init_rc=info[0], opt=None, score=None, transcript=None, origin_idx=None,
               mutant_idx=None, tb_null=None, would_panick=None)
    if info[0] != 0:
        return out
    out['num_rows'] = info[3]
    if info[4] < 0:
        out['opt'] = (-1, -1)
        return out
    out['opt'] = (info[4], info[5])
    out['score'] = score.value
    st = info[9]
    out['would_panick'] = bool(st & 4)
    out['tb_null'] = bool(st & 2) and not (st & 4)
    out['badpath'] = bool(st & 8)
    if not (st & 4) and not (st & 2):
        out['transcript'] = txbuf.value.decode('ascii')
        out['origin_idx'] = info[6]
        out['mutant_idx'] = info[7]
    return out
