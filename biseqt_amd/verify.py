"""Size-independent checks of alignment results (used by the GPU tests and by ``bench.py`` on the batch it times).

The reference accumulates a choice's score along its actual ``base`` chain (``biseqt/pwlib/_pw_internals.c:232,
268-278``), so re-scoring a transcript the way ``Alignment.calculate_score`` does (``biseqt/pw.py:391-428``: a
substitution score per M/S, ``ge`` per gap op plus ``go`` for every maximal run of one gap op) must reproduce the
reported score, every M/S must agree with the letters, and the path must end in the reported end cell.  These
hold for any correct result at any size, so they are what a 10 000-pair batch or a 100 kb pair is checked by.
"""
import numpy as np


def rescore(origin, mutant, transcript, origin_idx, mutant_idx, match, mismatch, go, ge):
    """Re-score one transcript (str / bytes / uint8 array) from its start cell; vectorised.

    Returns ``(score, end_x, end_y, letters_ok)`` with ``letters_ok`` False if some M sits on unequal letters
    or some S on equal ones (or the path leaves the sequences)."""
    if isinstance(transcript, str):
        transcript = transcript.encode('ascii')
    ops = np.frombuffer(transcript, dtype=np.uint8) if not isinstance(transcript, np.ndarray) else transcript
    n = len(ops)
    is_m, is_s, is_d, is_i = ops == 77, ops == 83, ops == 68, ops == 73
    if not (is_m | is_s | is_d | is_i).all():
        return None, -1, -1, False
    diag = is_m | is_s
    dx = (diag | is_d).astype(np.int64)
    dy = (diag | is_i).astype(np.int64)
    x = origin_idx + np.cumsum(dx) - dx          # coordinates BEFORE each op
    y = mutant_idx + np.cumsum(dy) - dy
    ex, ey = int(origin_idx + dx.sum()), int(mutant_idx + dy.sum())
    if ex > len(origin) or ey > len(mutant):
        return None, ex, ey, False
    o, m = np.asarray(origin), np.asarray(mutant)
    eq = o[x[diag]] == m[y[diag]]
    letters_ok = bool((eq == is_m[diag]).all())
    gap = ~diag
    prev = np.concatenate([[0], ops[:-1]]) if n else ops
    opens = gap & (ops != prev)
    score = match * int(is_m.sum()) + mismatch * int(is_s.sum()) + ge * int(gap.sum()) + go * int(opens.sum())
    return score, ex, ey, letters_ok


def end_cell_xy(opt_i, opt_j, banded, dmin=0):
    """Table coordinates as ``dptable_solve`` returns them -> (x, y) (``_xy_from_cellpos``,
    ``_pw_internals.c:100-114``)."""
    if not banded:
        return int(opt_i), int(opt_j)
    d = int(opt_i) + int(dmin)
    return int(opt_j) + max(d, 0), int(opt_j) - min(d, 0)


def check_batch(origins, mutants, results, transcripts, match, mismatch, go, ge, banded=False, dmins=None):
    """All pairs of a batch: returns the list of indices that violate a property (empty = all good).
    Pairs without an alignment (``opt_i < 0``) or whose traceback is empty / would panic are skipped."""
    bad = []
    for k in range(len(origins)):
        r = results[k]
        if r['opt_i'] < 0:
            continue
        if (r['status'] & 6) or transcripts[k] is None:
            continue
        s, ex, ey, ok = rescore(origins[k], mutants[k], transcripts[k], int(r['origin_idx']), int(r['mutant_idx']),
                                match, mismatch, go, ge)
        exy = end_cell_xy(r['opt_i'], r['opt_j'], banded, 0 if dmins is None else dmins[k])
        if not ok or s != r['score'] or (ex, ey) != exy:
            bad.append(k)
    return bad


# ---- many alignments at once: packed transcripts, reads in one arena, several host processes ---------------------------
def _check_packed_range(args):
    """Worker of :func:`check_packed_parallel`: pairs [k0, k1) of a batch whose reads live in a shared-memory arena and
    whose transcripts are packed back to back in another.  Returns (checked, list of bad pair indices)."""
    from multiprocessing import shared_memory
    (arena_name, arena_n, tx_name, tx_n, k0, k1, roff, rlen, pidx, rec, off, lo, hi, scores, overlap) = args
    sa, st = shared_memory.SharedMemory(name=arena_name), shared_memory.SharedMemory(name=tx_name)
    try:
        arena = np.ndarray((arena_n,), np.uint8, buffer=sa.buf)
        txb = np.ndarray((tx_n,), np.uint8, buffer=st.buf)
        match, mismatch, go, ge = scores
        bad, checked = [], 0
        for k in range(k0, k1):
            r = rec[k - k0]
            if r['opt_i'] < 0 or (r['status'] & 6) or r['tx_len'] <= 0:
                bad.append(k)                                   # (every pair of such a job has an alignment)
                continue
            a, c = pidx[k - k0]
            o = arena[roff[a]:roff[a] + rlen[a]]
            m = arena[roff[c]:roff[c] + rlen[c]]
            ops = txb[off[k - k0]:off[k - k0 + 1]]
            x0, y0 = int(r['origin_idx']), int(r['mutant_idx'])
            s, ex, ey, ok = rescore(o, m, ops, x0, y0, match, mismatch, go, ge)
            exy = end_cell_xy(r['opt_i'], r['opt_j'], True, lo[k - k0])
            good = ok and s == r['score'] and (ex, ey) == exy
            if good and overlap:
                # an overlap alignment starts on the table edge and ends on the last row or column, inside its band
                d = x0 - y0 + np.cumsum((ops == 68).astype(np.int64) - (ops == 73).astype(np.int64))
                good = (x0 == 0 or y0 == 0) and (ex == len(o) or ey == len(m)) and \
                    lo[k - k0] <= min(int(d.min()), x0 - y0) and max(int(d.max()), x0 - y0) <= hi[k - k0]
            if not good:
                bad.append(k)
            checked += 1
        return checked, bad
    finally:
        sa.close(); st.close()


def check_packed_parallel(pool, arena_shm, arena_n, roff, rlen, pidx, records, packed, offsets, lo, hi, scores, overlap=True,
                          chunk=4000):
    """Re-scores every alignment of a batch on the host cores (``pool``: a multiprocessing pool of spawned workers): the reads
    sit in the shared-memory block ``arena_shm`` (read ``a`` at ``roff[a]``, ``rlen[a]`` letters), the batch's transcripts come
    packed (``BatchAligner.packed``).  Returns (checked, bad indices)."""
    from multiprocessing import shared_memory
    n = len(pidx)
    st = shared_memory.SharedMemory(create=True, size=max(int(len(packed)), 1))
    try:
        np.ndarray((len(packed),), np.uint8, buffer=st.buf)[:] = packed
        jobs = []
        for k0 in range(0, n, chunk):
            k1 = min(n, k0 + chunk)
            jobs.append((arena_shm.name, arena_n, st.name, len(packed), k0, k1, roff, rlen, pidx[k0:k1], records[k0:k1],
                         np.asarray(offsets[k0:k1 + 1], np.int64), np.asarray(lo[k0:k1]), np.asarray(hi[k0:k1]), scores, overlap))
        checked, bad = 0, []
        for c, b in pool.imap_unordered(_check_packed_range, jobs):
            checked += c
            bad += b
        return checked, sorted(bad)
    finally:
        st.close(); st.unlink()
