# -*- coding: utf-8 -*-
"""Pairwise alignment: the ``Aligner`` / ``Alignment`` API of the reference's ``biseqt/pw.py`` on top
of the MI355X library (``biseqt_amd/pwlib/pwlib.so``).

    >>> from biseqt_amd.sequence import Alphabet
    >>> from biseqt_amd.pw import Aligner
    >>> A = Alphabet('ACGT')
    >>> S, T = A.parse('AAACGCGT'), A.parse('AACGCCTT')
    >>> with Aligner(S, T) as aligner:
    ...     score = aligner.solve()          # 6.0
    ...     aln = aligner.traceback()        # transcript MMDMMMIDMI
    >>> print(aln)
    origin[0]: AAACGC-GT-
    mutant[0]: AA-CGCC-TT

Same names, keyword arguments, defaults, assertions and return conventions as the reference
(``pw.py:119-319`` Aligner, ``:322-572`` Alignment); the C structs are built the same way
(``pw.py:203-245``) and handed to the same four C functions -- which here run the DP on the GPU.
Binding is ctypes instead of cffi (cffi is not installed in this image; the C ABI is unchanged, see
INTEGRATION.md).  For many pairs at once use :mod:`biseqt_amd.batch`.
"""
import ctypes as C
import re
from collections import namedtuple

from . import _pwlib as W
from .sequence import Sequence

lib = W.load()
"""The loaded shared object; fails at import time if it has not been built (no CPU fallback)."""

# alignment modes (reference pw.py:71-78)
STD_MODE = W.STD_MODE
BANDED_MODE = W.BANDED_MODE
# standard alignment types (pw.py:80-100)
GLOBAL = W.GLOBAL
LOCAL = W.LOCAL
START_ANCHORED = W.START_ANCHORED
END_ANCHORED = W.END_ANCHORED
OVERLAP = W.OVERLAP
START_ANCHORED_OVERLAP = W.START_ANCHORED_OVERLAP
END_ANCHORED_OVERLAP = W.END_ANCHORED_OVERLAP
# banded alignment types (pw.py:102-110)
B_GLOBAL = W.B_GLOBAL
B_OVERLAP = W.B_OVERLAP
B_LOCAL = W.B_LOCAL

ALN_TYPES = {
    STD_MODE: [GLOBAL, LOCAL, START_ANCHORED, END_ANCHORED, OVERLAP,
               START_ANCHORED_OVERLAP, END_ANCHORED_OVERLAP],
    BANDED_MODE: [B_GLOBAL, B_OVERLAP, B_LOCAL],
}

# minimal stand-in for termcolor (not installed here): same escape codes termcolor.colored emits
_ANSI = {'green': '\033[32m', 'red': '\033[31m'}
_RESET = '\033[0m'


def _colored(text, color=None, on_color=None):
    if color is None:
        return text
    return _ANSI[color] + text + _RESET


class Aligner(object):
    """A context that solves one pairwise alignment problem on the GPU.

    Memory (host table skeleton + device buffers) is allocated on entering the context and released
    on leaving it; ``solve`` and ``traceback`` are called explicitly inside it (reference
    ``pw.py:119-163`` documents the keyword arguments: ``origin_range``, ``mutant_range``,
    ``alnmode``, ``alntype``, ``subst_scores``, ``match_score``, ``mismatch_score``, ``go_score``,
    ``ge_score``, ``max_new_mins``, ``diag_range``, ``min_score``).
    """

    def __init__(self, origin, mutant, **kw):
        self.min_score = kw.get('min_score', float('-inf'))
        self.alnmode = kw.get('alnmode', STD_MODE)
        self.alntype = kw.get('alntype', GLOBAL)
        assert self.alnmode in [STD_MODE, BANDED_MODE]
        assert self.alntype in ALN_TYPES[self.alnmode]

        assert isinstance(origin, Sequence) and isinstance(mutant, Sequence)
        assert origin.alphabet == mutant.alphabet
        self.origin, self.mutant = origin, mutant
        self.alphabet = origin.alphabet

        origin_range = kw.get('origin_range', (0, len(self.origin)))
        mutant_range = kw.get('mutant_range', (0, len(self.mutant)))
        assert 0 <= origin_range[0] <= origin_range[1] <= len(self.origin)
        assert 0 <= mutant_range[0] <= mutant_range[1] <= len(self.mutant)
        self.origin_range, self.mutant_range = origin_range, mutant_range

        self.go_score = kw.get('go_score', 0)
        self.ge_score = kw.get('ge_score', 0)
        L = len(self.alphabet)
        subst_scores = kw.get('subst_scores', None)
        if subst_scores is None:
            mismatch = kw.get('mismatch_score', 0)
            match = kw.get('match_score', 1)
            subst_scores = [[match if i == j else mismatch for i in range(L)] for j in range(L)]
        assert isinstance(subst_scores, list) and len(subst_scores) == L
        self.subst_scores = subst_scores

        self.max_new_mins = kw.get('max_new_mins', -1)
        self.diag_range = kw.get('diag_range', None)

        # the C data structures, laid out as pw.py:203-245 lays them out
        self.c_subst_scores_rows = [(C.c_double * L)(*[float(v) for v in self.subst_scores[i]])
                                    for i in range(L)]
        self.c_subst_scores = (C.POINTER(C.c_double) * L)(
            *[C.cast(r, C.POINTER(C.c_double)) for r in self.c_subst_scores_rows])
        self.c_alnscores = W.alnscores(
            C.cast(self.c_subst_scores, C.POINTER(C.POINTER(C.c_double))),
            float(self.go_score), float(self.ge_score))
        self.c_origin = (C.c_int * max(1, len(self.origin)))(*self.origin.contents)
        self.c_mutant = (C.c_int * max(1, len(self.mutant)))(*self.mutant.contents)
        self.c_alnframe = W.alnframe(
            C.cast(self.c_origin, C.POINTER(C.c_int)), C.cast(self.c_mutant, C.POINTER(C.c_int)),
            W.intpair(*self.origin_range), W.intpair(*self.mutant_range))
        if self.alnmode == STD_MODE:
            self.c_alnparams = W.std_alnparams(self.alntype)
        elif self.alnmode == BANDED_MODE:
            self.min_diag, self.max_diag = kw['diag_range']
            assert -len(mutant) <= self.min_diag <= self.max_diag <= len(origin)
            self.c_alnparams = W.banded_alnparams(self.alntype, self.min_diag, self.max_diag)
        self.c_alnprob = W.alnprob(
            C.pointer(self.c_alnframe), C.pointer(self.c_alnscores), self.max_new_mins,
            self.alnmode, C.cast(C.pointer(self.c_alnparams), C.c_void_p))
        self.c_dptable = W.dptable(None, -1, None, C.pointer(self.c_alnprob))
        self.opt = None

    def __enter__(self):
        """Allocates the table skeleton (dimensions, band clamp and feasibility as the reference)."""
        if lib.dptable_init(C.byref(self.c_dptable)) == -1:
            raise Exception('Failed to initialize the DP table.')
        return self

    def __exit__(self, *args):
        """Frees host and device memory of the table."""
        lib.dptable_free(C.byref(self.c_dptable))

    def solve(self):
        """Fills the table (on the GPU) and reports the optimal score, or None if there is no
        alignment or it scores below ``min_score`` (reference ``pw.py:259-276``)."""
        opt = lib.dptable_solve(C.byref(self.c_dptable))
        self.opt = opt
        if self.opt.i == -1 or self.opt.j == -1:
            self.opt = None
            return None
        score = self.c_dptable.cells[self.opt.i][self.opt.j].choices[0].score
        if score < self.min_score:
            self.opt = None
            return None
        return score

    def table_scores(self):
        """2D list of the scores calculated by :func:`solve` (standard mode only), indexed like the
        reference's (``pw.py:278-285``): rows ``origin_range``, columns ``mutant_range``."""
        if self.alnmode != STD_MODE:
            raise NotImplementedError
        cells = self.c_dptable.cells
        return [[cells[i][j].choices[0].score for j in range(*self.mutant_range)]
                for i in range(*self.origin_range)]

    def traceback(self):
        """Traces back the optimal alignment found by :func:`solve`; an :class:`Alignment` or None
        (reference ``pw.py:287-306``)."""
        if self.opt is None:
            return None
        alignment = lib.dptable_traceback(C.byref(self.c_dptable), self.opt)
        assert bool(alignment)
        transcript = alignment.contents.transcript.decode('ascii')
        if not transcript:
            return None
        return Alignment(self.origin, self.mutant, transcript,
                         score=alignment.contents.score,
                         origin_start=alignment.contents.origin_idx,
                         mutant_start=alignment.contents.mutant_idx)

    def calculate_score(self, alignment):
        """Scores a given alignment with this aligner's scores (``pw.py:308-319``)."""
        return alignment.calculate_score(self.subst_scores, self.go_score, self.ge_score)


class Alignment(object):
    """A pairwise alignment: two sequences, an edit transcript over ``M``, ``S``, ``I``, ``D`` and the
    positions where it starts (reference ``pw.py:322-365``)."""

    def __init__(self, origin, mutant, transcript, score=None, origin_start=0, mutant_start=0):
        assert isinstance(origin, Sequence) and isinstance(mutant, Sequence)
        assert origin.alphabet == mutant.alphabet
        self.alphabet = origin.alphabet
        transcript = str(transcript)
        # (counted with str.count, not a Python loop: transcripts of long alignments have 10^4 - 10^5 ops)
        assert sum(transcript.count(c) for c in 'MSID') == len(transcript)
        assert len(transcript) > 0
        origin_end = origin_start + self.projected_len(transcript, on='origin')
        mutant_end = mutant_start + self.projected_len(transcript, on='mutant')
        assert 0 <= origin_start and origin_end <= len(origin)
        assert 0 <= mutant_start and mutant_end <= len(mutant)
        self.transcript = str(transcript)
        self.origin, self.mutant = origin, mutant
        self.origin_start, self.mutant_start = origin_start, mutant_start
        self.score = score

    def __str__(self):
        return self.render_term(term_width=float('+inf'), margin=0, colored=0)

    def __eq__(self, other):
        # the score is not part of equality (pw.py:359-365)
        assert isinstance(other, Alignment)
        return other.origin == self.origin and \
            other.mutant == self.mutant and \
            other.transcript == self.transcript and \
            other.origin_start == self.origin_start and \
            other.mutant_start == self.mutant_start

    def __ne__(self, other):
        return not self.__eq__(other)

    __hash__ = None

    @classmethod
    def projected_len(cls, transcript, on='origin'):
        """Letters of ``origin`` (ops M, S, D) or ``mutant`` (ops M, S, I) a transcript covers
        (``pw.py:367-389``)."""
        assert on in ['origin', 'mutant']
        ops = 'MSD' if on == 'origin' else 'MSI'
        if isinstance(transcript, str):
            return sum(transcript.count(op) for op in ops)
        return sum(int(op in ops) for op in transcript)

    def calculate_score(self, subst_scores, go_score, ge_score):
        """Re-scores the transcript: substitution scores for M/S, ``go + ge * n`` for every maximal
        run of n equal gap ops (``pw.py:391-428``)."""
        score = 0.
        i, j = self.origin_start, self.mutant_start
        for run in re.finditer(r'(.)\1*', self.transcript):
            op, num = run.group(1), len(run.group(0))
            if op in 'MS':
                score += sum(subst_scores[self.origin[i + k]][self.mutant[j + k]]
                             for k in range(num))
                i, j = i + num, j + num
            else:
                assert op in 'ID'
                score += go_score + ge_score * num
                if op == 'I':
                    j = j + num
                else:
                    i = i + num
        return score

    def truncate_to_match(self):
        """The sub-alignment between the first and the last ``M``, or None (``pw.py:430-448``)."""
        origin_start, mutant_start = self.origin_start, self.mutant_start
        tx_start, tx_end = self.transcript.find('M'), self.transcript.rfind('M')
        if tx_start < 0:
            # the reference's loop runs off the end of the transcript here (IndexError, pw.py:436): same failure
            raise IndexError('string index out of range')
        head = self.transcript[:tx_start]
        origin_start += head.count('D') + head.count('S')
        mutant_start += head.count('I') + head.count('S')
        if tx_start < tx_end:
            return Alignment(self.origin, self.mutant, self.transcript[tx_start:tx_end + 1],
                             origin_start=origin_start, mutant_start=mutant_start)
        return None

    def render_term(self, term_width=120, margin=0, colored=True):
        """Two-line textual rendering, wrapped at ``term_width`` (>= 30), with ``margin`` letters of
        context around the aligned region and optional ANSI colours (green matches, red
        substitutions); behaviour of ``pw.py:450-572``."""
        assert term_width >= 30
        assert margin >= 0
        letlen = self.alphabet._letlen
        Carriage = namedtuple('carriage', ['pos', 'o_idx', 'm_idx', 'o_line', 'm_line'])
        op_color = {'M': 'green', 'S': 'red'}

        def start_line(o_idx, m_idx):
            o_line, m_line = 'origin[%d]: ' % o_idx, 'mutant[%d]: ' % m_idx
            pos = max(len(o_line), len(m_line))
            assert pos <= term_width, 'Alignment preamble does not fit in width %d' % term_width
            return Carriage(pos=pos, o_idx=o_idx, m_idx=m_idx, o_line=o_line, m_line=m_line)

        def flush(car):
            width = max(len(car.o_line), len(car.m_line))
            return '%s\n%s\n' % (car.o_line.rjust(width), car.m_line.rjust(width))

        def forward(car, op=None):
            gap = '.' * letlen if op is None else '-' * letlen
            o_txt, m_txt = gap, gap
            if op is None:      # margin: plain letters where the sequences have them
                if 0 <= car.o_idx < len(self.origin):
                    o_txt = self.alphabet[self.origin[car.o_idx]]
                if 0 <= car.m_idx < len(self.mutant):
                    m_txt = self.alphabet[self.mutant[car.m_idx]]
            else:
                assert op in 'MSID'
                if op in 'MSD':
                    o_txt = self.alphabet[self.origin[car.o_idx]]
                if op in 'MSI':
                    m_txt = self.alphabet[self.mutant[car.m_idx]]
            length = len(o_txt)
            assert length == len(m_txt)
            color = op_color.get(op) if colored else None
            o_txt, m_txt = _colored(o_txt, color), _colored(m_txt, color)
            out = ''
            if car.pos >= term_width:
                out += flush(car)
                car = start_line(car.o_idx, car.m_idx)
            return out, Carriage(pos=car.pos + length,
                                 o_idx=car.o_idx + int(op is None or op in 'MSD'),
                                 m_idx=car.m_idx + int(op is None or op in 'MSI'),
                                 o_line=car.o_line + o_txt, m_line=car.m_line + m_txt)

        pre = min(margin, max(self.origin_start, self.mutant_start) * letlen)
        car = start_line(self.origin_start - pre, self.mutant_start - pre)
        output = ''
        for _ in range(pre):
            out, car = forward(car)
            output += out
        for op in self.transcript:
            out, car = forward(car, op=op)
            output += out
        post = min(margin, max((len(self.origin) - car.o_idx) * letlen,
                               (len(self.mutant) - car.m_idx) * letlen))
        for _ in range(post):
            out, car = forward(car)
            output += out
        return output + flush(car)
