"""ctypes driver for the pwlib C ABI -- TEST INFRASTRUCTURE ONLY.

Drives any shared object that exports the reference's four entry points
(`dptable_init`, `dptable_solve`, `dptable_traceback`, `dptable_free`;
reference `biseqt/pwlib/pwlib.h:204-242`) the way the reference's own caller does
(`biseqt/pw.py:203-306`): build the problem structs, init, solve, read
`cells[opt.i][opt.j].choices[0].score` straight out of C memory, trace back, free.

It is used for three things, all of them checking, never product:
  * driving `oracle/_ref/pwlib_ref.so` (the reference compiled from /root/reference by
    oracle/Makefile) to generate the golden vectors under tests/golden/ and to pin
    oracle/pw_oracle.c;
  * timing that same library as bench.py's `cpu_baseline` (kind "reference");
  * driving this repo's own libpwlib.so through the *identical* code path in the ABI tests.

Struct layouts follow `pwlib.h:15-187` under the LP64 C ABI (sizeof: intpair 8, alnscores 24,
alnframe 32, std_alnparams 4, banded_alnparams 12, alnprob 32, alnchoice 32, dpcell 16,
dptable 32, alignment 24).
"""
import ctypes as C
import os

STD_MODE, BANDED_MODE = 0, 1
GLOBAL, LOCAL, START_ANCHORED, END_ANCHORED, OVERLAP, START_ANCHORED_OVERLAP, \
    END_ANCHORED_OVERLAP = range(7)
B_GLOBAL, B_LOCAL, B_OVERLAP = range(3)

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SO = os.path.join(HERE, '_ref', 'pwlib_ref.so')
REF_SO_O2 = os.path.join(HERE, '_ref', 'pwlib_ref_O2.so')


class intpair(C.Structure):
    _fields_ = [('i', C.c_int), ('j', C.c_int)]


class alnscores(C.Structure):
    _fields_ = [('subst_scores', C.POINTER(C.POINTER(C.c_double))),
                ('gap_open_score', C.c_double),
                ('gap_extend_score', C.c_double)]


class alnframe(C.Structure):
    _fields_ = [('origin', C.POINTER(C.c_int)),
                ('mutant', C.POINTER(C.c_int)),
                ('origin_range', intpair),
                ('mutant_range', intpair)]


class std_alnparams(C.Structure):
    _fields_ = [('type', C.c_int)]


class banded_alnparams(C.Structure):
    _fields_ = [('type', C.c_int), ('dmin', C.c_int), ('dmax', C.c_int)]


class alnprob(C.Structure):
    _fields_ = [('frame', C.POINTER(alnframe)),
                ('scores', C.POINTER(alnscores)),
                ('max_new_mins', C.c_int),
                ('mode', C.c_int),
                ('params', C.c_void_p)]   # union {std_alnparams*, banded_alnparams*}


class alnchoice(C.Structure):
    pass


alnchoice._fields_ = [('op', C.c_char),
                      ('score', C.c_double),
                      ('base', C.POINTER(alnchoice)),
                      ('mins_cd', C.c_int),
                      ('cur_min', C.c_int)]


class dpcell(C.Structure):
    _fields_ = [('num_choices', C.c_int),
                ('choices', C.POINTER(alnchoice))]


class dptable(C.Structure):
    _fields_ = [('cells', C.POINTER(C.POINTER(dpcell))),
                ('num_rows', C.c_int),
                ('row_lens', C.POINTER(C.c_int)),
                ('prob', C.POINTER(alnprob))]


class alignment(C.Structure):
    _fields_ = [('origin_idx', C.c_int),
                ('mutant_idx', C.c_int),
                ('score', C.c_double),
                ('transcript', C.c_char_p)]


SIZEOF = dict(intpair=8, alnscores=24, alnframe=32, std_alnparams=4, banded_alnparams=12,
              alnprob=32, alnchoice=32, dpcell=16, dptable=32, alignment=24)


def check_layout():
    for name, size in SIZEOF.items():
        assert C.sizeof(globals()[name]) == size, (name, C.sizeof(globals()[name]), size)


_libs = {}


def load(path=REF_SO):
    """dlopen a pwlib-ABI shared object and declare the four prototypes."""
    path = os.path.abspath(path)
    if path in _libs:
        return _libs[path]
    lib = C.CDLL(path)
    lib.dptable_init.argtypes = [C.POINTER(dptable)]
    lib.dptable_init.restype = C.c_int
    lib.dptable_solve.argtypes = [C.POINTER(dptable)]
    lib.dptable_solve.restype = intpair
    lib.dptable_traceback.argtypes = [C.POINTER(dptable), intpair]
    lib.dptable_traceback.restype = C.POINTER(alignment)
    lib.dptable_free.argtypes = [C.POINTER(dptable)]
    lib.dptable_free.restype = None
    _libs[path] = lib
    return lib


class Problem(object):
    """One alignment problem laid out exactly as `pw.py:203-245` lays it out."""

    def __init__(self, origin, mutant, mode=STD_MODE, alntype=GLOBAL, subst=None, L=None,
                 match=1., mismatch=0., go=0., ge=0., diag_range=None,
                 origin_range=None, mutant_range=None, max_new_mins=-1):
        origin = [int(c) for c in origin]
        mutant = [int(c) for c in mutant]
        if L is None:
            L = (len(subst) if subst is not None else max(origin + mutant + [0]) + 1)
        if subst is None:
            # pw.py:192-195
            subst = [[match if i == j else mismatch for i in range(L)] for j in range(L)]
        self.L = L
        self.subst = subst
        self.mode, self.alntype = mode, alntype
        self._rows = [(C.c_double * L)(*[float(v) for v in subst[i]]) for i in range(L)]
        self._rowptrs = (C.POINTER(C.c_double) * L)(
            *[C.cast(r, C.POINTER(C.c_double)) for r in self._rows])
        self.scores = alnscores(C.cast(self._rowptrs, C.POINTER(C.POINTER(C.c_double))),
                                float(go), float(ge))
        self._origin = (C.c_int * max(1, len(origin)))(*origin)
        self._mutant = (C.c_int * max(1, len(mutant)))(*mutant)
        orange = origin_range if origin_range is not None else (0, len(origin))
        mrange = mutant_range if mutant_range is not None else (0, len(mutant))
        self.frame = alnframe(C.cast(self._origin, C.POINTER(C.c_int)),
                              C.cast(self._mutant, C.POINTER(C.c_int)),
                              intpair(*orange), intpair(*mrange))
        if mode == STD_MODE:
            self.params = std_alnparams(alntype)
        else:
            self.params = banded_alnparams(alntype, diag_range[0], diag_range[1])
        self.prob = alnprob(C.pointer(self.frame), C.pointer(self.scores), max_new_mins, mode,
                            C.cast(C.pointer(self.params), C.c_void_p))
        self.table = dptable(None, -1, None, C.pointer(self.prob))


def run(lib, P, want_table=False, do_traceback=True, end=None):
    """init -> solve -> (traceback) -> free.  Returns a plain dict.

    NB the reference calls exit(1) from dptable_traceback when the path starts at cell (0,0) and
    holds no M/S (`pw.c:132-134`); callers driving the *reference* must filter such inputs first
    (oracle/pw_oracle.c predicts them: `would_panick`).
    """
    out = dict(init_rc=None, opt=None, score=None, transcript=None, origin_idx=None,
               mutant_idx=None, tb_null=None)
    T = P.table
    rc = lib.dptable_init(C.byref(T))
    out['init_rc'] = rc
    if P.mode == BANDED_MODE:
        out['band'] = (P.params.dmin, P.params.dmax)   # possibly clamped by the library
    if rc != 0:
        return out
    out['num_rows'] = T.num_rows
    opt = lib.dptable_solve(C.byref(T))
    out['opt'] = (opt.i, opt.j)
    if opt.i != -1 and opt.j != -1:
        out['score'] = T.cells[opt.i][opt.j].choices[0].score    # pw.py:272
        if want_table and P.mode == STD_MODE:
            X = P.frame.origin_range.j - P.frame.origin_range.i
            Y = P.frame.mutant_range.j - P.frame.mutant_range.i
            out['table'] = [[(T.cells[i][j].choices[0].score if T.cells[i][j].num_choices > 0
                              else None) for j in range(Y + 1)] for i in range(X + 1)]
        if do_traceback:
            e = opt if end is None else intpair(*end)
            aln = lib.dptable_traceback(C.byref(T), e)
            if not aln:
                out['tb_null'] = True
            else:
                out['tb_null'] = False
                out['transcript'] = aln.contents.transcript.decode('ascii')
                out['origin_idx'] = aln.contents.origin_idx
                out['mutant_idx'] = aln.contents.mutant_idx
                out['tb_score'] = aln.contents.score
    lib.dptable_free(C.byref(T))
    return out
