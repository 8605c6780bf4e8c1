"""ctypes wrapper around oracle/libpw_oracle.so (oracle/pw_oracle.c) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (biseqt_amd) never does.  Results come back as a dict with the same keys
`oracle/ref_driver.run` produces for the compiled reference, so the two can be compared directly.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, 'libpw_oracle.so')

STD_MODE, BANDED_MODE = 0, 1
GLOBAL, LOCAL, START_ANCHORED, END_ANCHORED, OVERLAP, START_ANCHORED_OVERLAP, \
    END_ANCHORED_OVERLAP = range(7)
B_GLOBAL, B_LOCAL, B_OVERLAP = range(3)


class pwo_problem(C.Structure):
    _fields_ = [('mode', C.c_int), ('type', C.c_int), ('X', C.c_int), ('Y', C.c_int),
                ('origin', C.POINTER(C.c_int)), ('mutant', C.POINTER(C.c_int)),
                ('L', C.c_int), ('subst', C.POINTER(C.c_double)),
                ('go', C.c_double), ('ge', C.c_double),
                ('dmin', C.c_int), ('dmax', C.c_int), ('max_new_mins', C.c_int)]


class pwo_result(C.Structure):
    _fields_ = [('init_rc', C.c_int), ('dmin_c', C.c_int), ('dmax_c', C.c_int),
                ('num_rows', C.c_int), ('cells', C.c_longlong),
                ('opt_i', C.c_int), ('opt_j', C.c_int), ('score', C.c_double),
                ('would_panick', C.c_int), ('tb_null', C.c_int),
                ('origin_idx', C.c_int), ('mutant_idx', C.c_int),
                ('tx_len', C.c_int), ('transcript', C.POINTER(C.c_char)),
                ('maskrule_ok', C.c_int)]


_lib = None


def build():
    """(Re)build libpw_oracle.so (and oracle/_ref when /root/reference is present)."""
    subprocess.check_call(['make', '-s', '-C', HERE, 'all'])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO):
            build()
        _lib = C.CDLL(SO)
        _lib.pwo_solve.argtypes = [C.POINTER(pwo_problem), C.POINTER(pwo_result),
                                   C.POINTER(C.c_double), C.POINTER(C.c_ubyte)]
        _lib.pwo_solve.restype = C.c_int
        _lib.pwo_cells.argtypes = [C.POINTER(pwo_problem)]
        _lib.pwo_cells.restype = C.c_longlong
        _lib.pwo_free_result.argtypes = [C.POINTER(pwo_result)]
    return _lib


def _problem(origin, mutant, mode, alntype, subst, L, match, mismatch, go, ge, diag_range,
             origin_range, mutant_range, max_new_mins):
    o = np.ascontiguousarray(np.asarray(origin, dtype=np.int32))
    m = np.ascontiguousarray(np.asarray(mutant, dtype=np.int32))
    if L is None:
        L = len(subst) if subst is not None else int(max([0] + list(o) + list(m))) + 1
    if subst is None:
        subst = [[match if i == j else mismatch for i in range(L)] for j in range(L)]
    S = np.ascontiguousarray(np.asarray(subst, dtype=np.float64).reshape(L, L))
    orange = origin_range if origin_range is not None else (0, len(o))
    mrange = mutant_range if mutant_range is not None else (0, len(m))
    of = np.ascontiguousarray(o[orange[0]:orange[1]])
    mf = np.ascontiguousarray(m[mrange[0]:mrange[1]])
    if len(of) == 0:
        of = np.zeros(1, np.int32)
    if len(mf) == 0:
        mf = np.zeros(1, np.int32)
    dr = diag_range if diag_range is not None else (0, 0)
    P = pwo_problem(mode, alntype, orange[1] - orange[0], mrange[1] - mrange[0],
                    of.ctypes.data_as(C.POINTER(C.c_int)), mf.ctypes.data_as(C.POINTER(C.c_int)),
                    L, S.ctypes.data_as(C.POINTER(C.c_double)), float(go), float(ge),
                    int(dr[0]), int(dr[1]), max_new_mins)
    return P, (of, mf, S), orange, mrange


def cells(origin_len, mutant_len, mode=STD_MODE, alntype=GLOBAL, diag_range=None):
    """Cells the reference allocates for a table of this shape (SURVEY 8d metric definition)."""
    dr = diag_range if diag_range is not None else (0, 0)
    P = pwo_problem(mode, alntype, origin_len, mutant_len, None, None, 1, None, 0., 0.,
                    int(dr[0]), int(dr[1]), -1)
    return lib().pwo_cells(C.byref(P))


def solve(origin, mutant, mode=STD_MODE, alntype=GLOBAL, subst=None, L=None, match=1.,
          mismatch=0., go=0., ge=0., diag_range=None, origin_range=None, mutant_range=None,
          max_new_mins=-1, want_table=False):
    P, keep, orange, mrange = _problem(origin, mutant, mode, alntype, subst, L, match, mismatch,
                                       go, ge, diag_range, origin_range, mutant_range,
                                       max_new_mins)
    R = pwo_result()
    H = M = None
    Hp = Mp = None
    if want_table:
        n = lib().pwo_cells(C.byref(P))
        if n > 0:
            H = np.empty(n, np.float64)
            M = np.empty(n, np.uint8)
            Hp = H.ctypes.data_as(C.POINTER(C.c_double))
            Mp = M.ctypes.data_as(C.POINTER(C.c_ubyte))
    rc = lib().pwo_solve(C.byref(P), C.byref(R), Hp, Mp)
    if rc != 0:
        raise ValueError('unsupported problem (rc=%d)' % rc)
    out = dict(init_rc=R.init_rc, opt=None, score=None, transcript=None, origin_idx=None,
               mutant_idx=None, tb_null=None, would_panick=None, cells=R.cells)
    if mode == BANDED_MODE:
        out['band'] = (R.dmin_c, R.dmax_c)
    if R.init_rc != 0:
        return out
    out['num_rows'] = R.num_rows
    out['opt'] = (R.opt_i, R.opt_j)
    if R.opt_i != -1:
        out['score'] = R.score
        out['would_panick'] = bool(R.would_panick)
        out['tb_null'] = bool(R.tb_null)
        out['maskrule_ok'] = bool(R.maskrule_ok)
        if not R.would_panick and not R.tb_null:
            out['transcript'] = C.string_at(R.transcript, R.tx_len).decode('ascii')
            out['origin_idx'] = R.origin_idx + orange[0]
            out['mutant_idx'] = R.mutant_idx + mrange[0]
            out['tb_score'] = R.score
    if want_table:
        out['H'] = H
        out['mask'] = M
    lib().pwo_free_result(C.byref(R))
    return out
