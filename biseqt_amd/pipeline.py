"""Seeds -> segments -> banded DP extension, everything heavy on the GPU.

The reference has no library function for this; the flow is written out in its experiments
(``experiments/blot_stats.py:374-470``): Word-Blot finds the similar segments of a pair of sequences, every segment
is turned into a pair of sub-frames a few word lengths larger, and a banded global alignment of the two frames is
solved, traced back and truncated to its first / last match.  Here the frames of ALL segments are solved by one
batch (`BatchAligner`), i.e. a handful of kernel launches, and the arithmetic that decides the frames and bands is
the reference's, statement by statement.
"""
import numpy as np

from . import _pwlib as W
from .batch import BatchAligner
from .blot import WordBlot
from .pw import Alignment
from .sequence import Sequence


def segment_frame(seg, lenS, lenT, wordlen):
    """Frames ``(i_start, i_end), (j_start, j_end)`` and band radius for one similar segment
    ``((d_min, d_max), (a_min, a_max))`` -- ``experiments/blot_stats.py:438-453``."""
    (i_start, i_end), (j_start, j_end) = WordBlot.to_ij_coordinates_seg(seg)
    i_end = min(lenS - 1, i_end)                       # to_ij_coordinates_seg might overflow
    j_end = min(lenT - 1, j_end)
    start_shift = min(i_start, j_start, 2 * wordlen)   # allow longer alignments to be found
    end_shift = min(lenS - i_end, lenT - j_end, 2 * wordlen)
    i_start, j_start = i_start - start_shift, j_start - start_shift
    i_end, j_end = i_end + end_shift, j_end + end_shift
    n_s, n_t = i_end - i_start, j_end - j_start
    rad = (seg[0][1] - seg[0][0]) // 2
    rad = min(n_s, n_t, max(rad, abs(n_s - n_t) + 2))
    return (i_start, i_end), (j_start, j_end), rad


def extend_segments(S, T, segments, wordlen, device=0, **aligner_kw):
    """Banded alignment of the frames of all ``segments`` (dicts with a ``segment`` key, as
    ``WordBlot.similar_segments`` yields them) in one GPU batch.  ``aligner_kw`` are ``Aligner`` keywords
    (``alnmode`` / ``alntype`` default to banded global as in the reference's experiment; ``diag_range`` is set
    per segment).  Returns one dict per segment: ``frame``, ``diag_range``, ``score``, ``alignment`` (an
    :class:`Alignment` on the frame sequences, or None), ``truncated`` (``alignment.truncate_to_match()``) and ``kernel``
    (the batch's fill kernel and score type, for diagnostics)."""
    assert isinstance(S, Sequence) and isinstance(T, Sequence)
    kw = dict(alnmode=W.BANDED_MODE, alntype=W.B_GLOBAL)
    kw.update(aligner_kw)
    assert kw['alnmode'] == W.BANDED_MODE, 'segments are extended by banded alignments'
    s, t = S.as_array(np.uint8), T.as_array(np.uint8)
    frames, pairs, bands = [], [], []
    for rec in segments:
        fi, fj, rad = segment_frame(rec['segment'], len(S), len(T), wordlen)
        frames.append((fi, fj))
        pairs.append((s[fi[0]:fi[1]], t[fj[0]:fj[1]]))
        bands.append((-rad, rad))
    if not frames:
        return []
    kw.pop('diag_range', None)
    with BatchAligner(pairs, alphabet_len=len(S.alphabet), diag_range=bands, device=device, **kw) as b:
        res = b.run()
        txs = b.transcripts(res)
        kernel = (b.kernel_name, b.score_dtype)
    out = []
    for k, (fi, fj) in enumerate(frames):
        rec = {'frame': (fi, fj), 'diag_range': bands[k], 'score': None, 'alignment': None, 'truncated': None, 'kernel': kernel}
        if res['opt_i'][k] >= 0 and txs[k]:
            rec['score'] = float(res['score'][k])
            aln = Alignment(S[fi[0]:fi[1]], T[fj[0]:fj[1]], txs[k], score=rec['score'],
                            origin_start=int(res['origin_idx'][k]), mutant_start=int(res['mutant_idx'][k]))
            rec['alignment'] = aln
            rec['truncated'] = aln.truncate_to_match() if 'M' in txs[k] else None
        out.append(rec)
    return out


def local_homology_scan(S, T, K_min, p_min, wordlen, g_max=.3, sensitivity=.99, mask=(), device=0,
                        aligner_kw=None):
    """Word-Blot local-homology scan followed by banded DP extension of every similar segment (BASELINE config 5;
    ``experiments/blot_stats.py:362-470``).  Default scores are the experiment's: match ``1 / p_min - 1``,
    mismatch -1, gap extend -1, gap open 0, banded global.  Returns ``(segments, extensions)``."""
    if aligner_kw is None:
        aligner_kw = dict(match_score=1. / p_min - 1, mismatch_score=-1, ge_score=-1, go_score=0)
    wb = WordBlot(S, T, g_max=g_max, sensitivity=sensitivity, alphabet=S.alphabet, wordlen=wordlen, mask=list(mask),
                  device=device)
    try:
        segments = list(wb.similar_segments(K_min, p_min))
    finally:
        wb.close()
    return segments, extend_segments(S, T, segments, wordlen, device=device, **aligner_kw)
