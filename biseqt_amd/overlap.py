"""Overlap discovery for many read pairs: band selection in one GPU pass, then one banded overlap-alignment batch.

The reference scores every pair of reads with its own ``WordBlotOverlap`` object
(``experiments/blot_overlaps.py:262-272``, ``biseqt/blot.py:497-579``).  :func:`overlap_bands` returns, per pair, the
same dict ``highest_scoring_overlap_band()`` returns (``d_band``, ``p``, ``len``, ``score``; None without seeds), all
pairs scored by one call into ``include/pw_overlap.h``; :func:`overlap_alignments` then gives every band to the
banded overlap aligner (``B_OVERLAP``) in one batch -- BASELINE config 4's "seeding -> banded DP".
Pairs for which more than one diagonal could reach the same estimated match probability (where the reference's
answer depends on the order of its seeds table) are re-scored by the single-pair path, which reproduces that order.
One documented divergence: two reads with IDENTICAL content are scored here as two different sequences, whereas the
reference turns such a pair into a self comparison (``seeds.py:33``) and ignores its main diagonal.
"""
import ctypes as C

import numpy as np
from scipy.special import erfcinv

from . import _pwlib as W
from .batch import BatchAligner
from .blot import H1_moments, WordBlotOverlap
from .sequence import Alphabet, Sequence

BAND_DTYPE = np.dtype([('n_seeds', '<i8'), ('w_best', '<f8'), ('d_best', '<i4'), ('n_best', '<i4'),
                       ('r_best', '<i4'), ('len_best', '<i4'), ('band_best', '<i4'), ('tie', '<i4'),
                       ('d_first', '<i4'), ('n_first', '<i4'), ('r_first', '<i4'), ('len_first', '<i4'),
                       ('band_first', '<i4'), ('pad_', '<i4')])
assert BAND_DTYPE.itemsize == 64


def _arena(reads):
    arrs = [r.as_array(np.uint8) if isinstance(r, Sequence) else np.ascontiguousarray(r, np.uint8) for r in reads]
    offs = np.zeros(len(arrs) + 1, np.int64)
    offs[1:] = np.cumsum([len(a) for a in arrs])
    arena = np.concatenate(arrs) if arrs else np.zeros(0, np.uint8)
    return arena, offs, arrs


def _same(a, b):
    a = a.as_array(np.uint8) if isinstance(a, Sequence) else np.asarray(a)
    b = b.as_array(np.uint8) if isinstance(b, Sequence) else np.asarray(b)
    return len(a) == len(b) and bool((a == b).all())


def raw_bands(reads, pairs, wordlen, alphabet_len, g_max, sensitivity, device=0):
    """The device records (BAND_DTYPE) for ``pairs`` = list of (i, j) indices into ``reads``; also returns the
    device time in ms."""
    assert 0 < g_max < 1 and 0 < sensitivity < 1
    lib = W.load()
    arena, offs, arrs = _arena(reads)
    n = len(pairs)
    rp = (W.pw_read_pair * max(n, 1))()
    for q, (i, j) in enumerate(pairs):
        rp[q] = W.pw_read_pair(int(offs[i]), int(offs[j]), len(arrs[i]), len(arrs[j]))
    out = np.zeros(max(n, 1), BAND_DTYPE)
    # the reference's constants, computed as it computes them (blot.py:109, 134-136, 538)
    len_coeff = 2. / (2 - g_max)
    radius_coeff = erfcinv(1. - sensitivity) * np.sqrt(2 * g_max)
    word_p_null = (1. / alphabet_len) ** wordlen
    arena_c = np.ascontiguousarray(arena)
    rc = lib.pw_overlap_bands(device, arena_c.ctypes.data, arena_c.size, rp, n, alphabet_len, wordlen,
                              float(len_coeff), float(radius_coeff), float(word_p_null), out.ctypes.data)
    if rc != 0:
        raise RuntimeError('pw_overlap_bands failed: ' + (lib.pw_overlap_last_error() or b'').decode())
    return out[:n], lib.pw_overlap_last_ms()


def _result(d, rad, L, n_in_band, word_p, alphabet_len, wordlen):
    if not word_p > 0:                                   # blot.py:541-545
        p_hat = 0
    else:
        p_hat = np.exp(np.log(word_p) / wordlen)
    p_hat = min(p_hat, 1)
    rad = np.float64(rad)                                # the reference carries np.ceil's float
    res = {'d_band': (d - rad, d + rad), 'p': p_hat, 'len': int(L)}
    area = 2 * rad * L
    mu_H1, sd_H1 = H1_moments(alphabet_len, wordlen, area, L, p_hat)
    res['score'] = (n_in_band - mu_H1) / sd_H1
    return res


def _records_to_results(reads, pairs, recs, wordlen, alphabet, g_max, sensitivity, device):
    """Device records -> the reference's dicts (shared by the pair-list and the all-pairs paths)."""
    L = len(alphabet)
    out, n_fallback = [], 0
    for q, r in enumerate(recs):
        if r['n_seeds'] == 0:
            out.append(None)
        elif not r['w_best'] > 0:
            # every p is 0: max() keeps the first row of the table (blot.py:569)
            word_p = (r['n_first'] + 1 - (2 * int(r['r_first']) * int(r['len_first'])) * ((1. / L) ** wordlen)) / r['len_first']
            out.append(_result(int(r['d_first']), int(r['r_first']), int(r['len_first']), int(r['band_first']), word_p, L, wordlen))
        elif r['tie'] > 1 and not _same(reads[pairs[q][0]], reads[pairs[q][1]]):
            i, j = pairs[q]
            S = reads[i] if isinstance(reads[i], Sequence) else Sequence(alphabet, tuple(int(c) for c in reads[i]))
            T = reads[j] if isinstance(reads[j], Sequence) else Sequence(alphabet, tuple(int(c) for c in reads[j]))
            wb = WordBlotOverlap(S, T, g_max=g_max, sensitivity=sensitivity, alphabet=alphabet, wordlen=wordlen, device=device)
            out.append(wb.highest_scoring_overlap_band())
            wb.close()
            n_fallback += 1
        else:
            out.append(_result(int(r['d_best']), int(r['r_best']), int(r['len_best']), int(r['band_best']), float(r['w_best']), L, wordlen))
    return out, n_fallback


def overlap_bands(reads, pairs, wordlen, alphabet, g_max, sensitivity, device=0, stats=None):
    """``highest_scoring_overlap_band()`` of every pair ``(i, j)`` (``reads[i]`` as S, ``reads[j]`` as T)."""
    assert isinstance(alphabet, Alphabet)
    recs, ms = raw_bands(reads, pairs, wordlen, len(alphabet), g_max, sensitivity, device)
    out, n_fallback = _records_to_results(reads, pairs, recs, wordlen, alphabet, g_max, sensitivity, device)
    if stats is not None:
        stats.update(device_ms=ms, fallback_pairs=n_fallback, pairs=len(pairs))
    return out


def raw_all_pairs(reads, wordlen, alphabet_len, g_max, sensitivity, device=0, max_pairs=None, rank=0, world=1):
    """All pairs ``a < b`` of ``reads`` that share at least one seed, through ONE k-mer index over all reads:
    returns ``(pairs (n, 2) int32, records BAND_DTYPE, device ms)``.  With ``world > 1`` only the pairs whose smaller
    read index is ``rank`` modulo ``world`` (one process per GPU, no data-path collective)."""
    assert 0 < g_max < 1 and 0 < sensitivity < 1
    lib = W.load()
    arena, offs, arrs = _arena(reads)
    R = len(arrs)
    cap = int(max_pairs) if max_pairs is not None else min(R * (R - 1) // 2, 1 << 26)
    pa, pb = np.zeros(max(cap, 1), np.int32), np.zeros(max(cap, 1), np.int32)
    out = np.zeros(max(cap, 1), BAND_DTYPE)
    n_out = C.c_int64(0)
    roff = np.ascontiguousarray(offs[:-1].astype(np.uint64))
    rlen = np.ascontiguousarray(np.diff(offs).astype(np.int32))
    arena_c = np.ascontiguousarray(arena)
    rc = lib.pw_overlap_all_pairs(device, arena_c.ctypes.data, arena_c.size, roff.ctypes.data, rlen.ctypes.data, R, alphabet_len,
                                  wordlen, float(2. / (2 - g_max)), float(erfcinv(1. - sensitivity) * np.sqrt(2 * g_max)),
                                  float((1. / alphabet_len) ** wordlen), int(rank), int(world), cap, pa.ctypes.data, pb.ctypes.data, out.ctypes.data,
                                  C.byref(n_out))
    if rc != 0:
        raise RuntimeError('pw_overlap_all_pairs failed: ' + (lib.pw_overlap_last_error() or b'').decode())
    n = n_out.value
    return np.stack([pa[:n], pb[:n]], axis=1), out[:n], lib.pw_overlap_last_ms()


def overlap_all_pairs(reads, wordlen, alphabet, g_max, sensitivity, device=0, max_pairs=None, stats=None):
    """The reference's all-pairs loop (``experiments/blot_overlaps.py:262-272``) in one device pass: returns a dict
    ``{(a, b): highest_scoring_overlap_band()}`` for the pairs ``a < b`` that share a seed; for every other pair the
    reference's answer is None."""
    assert isinstance(alphabet, Alphabet)
    pairs, recs, ms = raw_all_pairs(reads, wordlen, len(alphabet), g_max, sensitivity, device, max_pairs)
    plist = [(int(a), int(b)) for a, b in pairs.tolist()]
    res, n_fallback = _records_to_results(reads, plist, recs, wordlen, alphabet, g_max, sensitivity, device)
    if stats is not None:
        stats.update(device_ms=ms, fallback_pairs=n_fallback, pairs=len(plist))
    return dict(zip(plist, res))


def aligned_batches(arena, offs, lens, pidx, dr, alphabet_len, device=0, max_cells=2 * 10 ** 10, flags=0, **kw):
    """Banded overlap alignment of the read pairs ``pidx`` (read indices into the packed ``arena``, :func:`pack_reads`)
    with bands ``dr``, in batches of at most ``max_cells`` cells.  The reads go to the device ONCE (``DeviceArena``) and
    every batch refers to that copy; while one batch runs on the device the next one is planned on the host
    (``pw_batch_create`` is host work: band clamps, kernel geometry, descriptors).  Yields ``(start, stop, batch)`` with
    the batch solved, traced back and synchronised; the batch is destroyed when the generator moves on."""
    from .batch import DeviceArena
    pidx = np.ascontiguousarray(pidx, np.int64).reshape(-1, 2)
    dr = np.ascontiguousarray(dr, np.int64).reshape(-1, 2)
    cells = (dr[:, 1] - dr[:, 0] + 1) * np.minimum(lens[pidx[:, 0]], lens[pidx[:, 1]]).astype(np.int64)
    bounds, start = [], 0
    csum = np.cumsum(cells)
    while start < len(pidx):
        base = csum[start - 1] if start else 0
        stop = max(start + 1, int(np.searchsorted(csum, base + max_cells, 'right')))
        bounds.append((start, stop)); start = stop

    def create(lo, hi):
        return BatchAligner.from_arena(arena, offs, lens, pidx[lo:hi], dr[lo:hi], device_arena=dev, alnmode=W.BANDED_MODE,
                                       alntype=W.B_OVERLAP, alphabet_len=alphabet_len, device=device, flags=flags, **kw)

    with DeviceArena(arena, device) as dev:
        cur = nxt = None
        try:
            nxt = create(*bounds[0]) if bounds else None
            for q, (lo, hi) in enumerate(bounds):
                cur, nxt = nxt, None
                cur.solve(); cur.traceback()                       # asynchronous: the device is busy from here ...
                if q + 1 < len(bounds):
                    nxt = create(*bounds[q + 1])                   # ... while the next batch is planned
                cur.sync()
                yield lo, hi, cur
                cur.close(); cur = None
        finally:
            for b in (cur, nxt):
                if b is not None:
                    b.close()


def overlap_alignments(reads, pairs, bands, alphabet, p_min=0., device=0, max_cells=2 * 10 ** 10, want_transcripts=True,
                       **aligner_kw):
    """Banded overlap alignment (``B_OVERLAP``) of every pair whose band has ``p >= p_min``; the ``diag_range`` is the
    band clamped to the table as ``Aligner`` requires (``pw.py:224-226``).  All reads are uploaded ONCE and the pairs
    refer to them (``BatchAligner.from_arena``); the pairs are solved in batches of at most ``max_cells`` cells
    (tie masks take 0.5-0.6 bytes per cell of HBM).  ``bands`` is a list aligned with ``pairs`` (dicts or None).
    Returns a list with one entry per pair: None, or dict(score, transcript, origin_start, mutant_start,
    diag_range)."""
    from .batch import pack_reads
    arena, offs, lens = pack_reads(reads)
    sel, dr = [], []
    for q, ((i, j), band) in enumerate(zip(pairs, bands)):
        if band is None or band['p'] < p_min:
            continue
        lo = max(int(band['d_band'][0]), -int(lens[j]))
        hi = min(int(band['d_band'][1]), int(lens[i]))
        if lo > hi:
            continue
        sel.append(q); dr.append((lo, hi))
    out = [None] * len(pairs)
    if not sel:
        return out
    kw = dict(match_score=1, mismatch_score=-3, go_score=-5, ge_score=-2)
    kw.update(aligner_kw)
    pidx = np.array([pairs[q] for q in sel], np.int64)
    dr = np.array(dr, np.int64)
    for start, stop, b in aligned_batches(arena, offs, lens, pidx, dr, len(alphabet), device=device, max_cells=max_cells, **kw):
        res = b.results()
        txs = b.transcripts(res) if want_transcripts else [None] * (stop - start)
        for k in range(stop - start):
            if res['opt_i'][k] < 0:
                continue
            out[sel[start + k]] = dict(score=float(res['score'][k]), transcript=txs[k], origin_start=int(res['origin_idx'][k]),
                                       mutant_start=int(res['mutant_idx'][k]), diag_range=(int(dr[start + k, 0]), int(dr[start + k, 1])))
    return out


def raw_bands_sharded(reads, pairs, wordlen, alphabet_len, g_max, sensitivity, rank, world, device=None, gather_device=None):
    """Config 4 across GPUs: pair q is scored by rank ``q mod world`` (no data-path collective), the 64-byte band
    records are gathered to rank 0 in pair order (``torch.distributed``: RCCL on GPUs).  Returns the full record
    array on rank 0, None elsewhere."""
    from .distributed import gather_struct, shard_indices
    mine = shard_indices(len(pairs), rank, world)
    recs, _ = raw_bands(reads, [pairs[q] for q in mine], wordlen, alphabet_len, g_max, sensitivity,
                        device=rank if device is None else device)
    return gather_struct(recs, len(pairs), rank, world, device=gather_device)


def raw_all_pairs_sharded(reads, wordlen, alphabet_len, g_max, sensitivity, rank, world, device=None, max_pairs=None):
    """Config 4 across GPUs: every rank indexes all reads (cheap) and joins / scores the pairs whose smaller read
    index is ``rank`` modulo ``world``; pair lists and 64-byte records are gathered to rank 0 (ragged byte gather over
    ``torch.distributed``) and merged in ascending (a, b) order.  Returns ``(pairs, records)`` on rank 0, None elsewhere."""
    import torch
    from .distributed import gather_bytes
    pairs, recs, _ = raw_all_pairs(reads, wordlen, alphabet_len, g_max, sensitivity, device=rank if device is None else device,
                                   max_pairs=max_pairs, rank=rank, world=world)
    blob = np.concatenate([np.ascontiguousarray(pairs, np.int32).view(np.uint8).reshape(-1), recs.view(np.uint8).reshape(-1)])
    got = gather_bytes(torch.from_numpy(blob.copy()), rank, world)
    if rank != 0:
        return None
    all_pairs, all_recs = [], []
    for t in got:
        raw = t.cpu().numpy()
        n = raw.size // (8 + BAND_DTYPE.itemsize)
        all_pairs.append(raw[:8 * n].view(np.int32).reshape(n, 2))
        all_recs.append(raw[8 * n:].view(BAND_DTYPE))
    pairs, recs = np.concatenate(all_pairs), np.concatenate(all_recs)
    order = np.lexsort((pairs[:, 1], pairs[:, 0]))
    return pairs[order], recs[order]
