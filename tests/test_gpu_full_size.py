"""BASELINE configs 4 and 5 at FULL size inside the asserted GPU suite (round 2 ran them at full size only in the micro
benchmarks): size-independent properties, every alignment re-scored on the host (`biseqt_amd.verify`: the reference
accumulates a score along its actual path, so re-scoring a transcript must reproduce it -- _pw_internals.c:232, 268-278;
pw.py:391-428), kernel choices pinned.  A few seconds of device time each; the host work (read generation, 1.2 million
re-scorings) is spread over spawned processes."""
import multiprocessing as mp
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _reads_chunk(args):
    """Reads [r0, r1) of the config-4 read set: positions drawn by the parent, the genome regenerated from its seed."""
    seed, G, read_len, starts, r0 = args
    from biseqt_amd import synth
    g = synth.rand_seqs(synth.rng_for(seed), 1, G)[0]
    rng = synth.rng_for(seed * 1000 + 1 + r0)
    return [synth.mutate(rng, g[s:s + read_len], .05, .025, .025) for s in starts]


def test_config4_full_size_properties():
    """50 000 reads of 5 kb at 25x coverage of a 10 Mb genome, every pair of reads (1.25e9) through one k-mer index
    (k = 16), the pairs with p >= 0.8 through banded overlap alignment (B_OVERLAP, 1 / -3 / -5 / -2), about 1.2 million
    alignments and 2.8e11 cells: recall >= 99 % of the true overlaps longer than 500 bases, and EVERY alignment re-scored
    and checked to be an overlap alignment inside its band."""
    from multiprocessing import shared_memory
    from biseqt_amd import synth, verify, _pwlib as W
    from biseqt_amd.batch import pack_reads
    from biseqt_amd.overlap import aligned_batches, raw_all_pairs
    R, read_len, cov, k, seed = 50000, 5000, 25, 16, 4
    G = R * read_len // cov
    starts = synth.rng_for(seed + 77).integers(0, G - read_len, R)
    ctx = mp.get_context('spawn')                          # never fork a process that has touched the GPU
    workers = min(16, os.cpu_count() or 4)
    with ctx.Pool(workers) as pool:
        step = 2500
        reads = [r for part in pool.map(_reads_chunk, [(seed, G, read_len, starts[r0:r0 + step], r0) for r0 in range(0, R, step)])
                 for r in part]
        assert len(reads) == R
        pairs, recs, ms = raw_all_pairs(reads, k, 4, .2, .9, max_pairs=1 << 23)
        w = recs['w_best']
        p = np.where(w > 0, np.exp(np.log(np.maximum(w, 1e-300)) / k), 0.0)
        sel = np.flatnonzero(p >= .8)
        ov = np.minimum(starts[pairs[:, 0]], starts[pairs[:, 1]]) + read_len - np.maximum(starts[pairs[:, 0]], starts[pairs[:, 1]])
        order = np.argsort(starts); ss = starts[order]
        true_total = int(sum(np.searchsorted(ss, ss[i] + read_len - 500, 'left') - i - 1 for i in range(R)))
        found = int((ov[sel] > 500).sum())
        assert true_total > 1000000
        assert found >= 0.99 * true_total, (found, true_total)
        assert int((ov[sel] <= 0).sum()) <= 0.001 * len(sel) + 2
        arena, offs, lens = pack_reads(reads)
        pidx = pairs[sel].astype(np.int64)
        lo = np.maximum(recs['d_best'][sel].astype(np.int64) - recs['r_best'][sel], -lens[pidx[:, 1]].astype(np.int64))
        hi = np.minimum(recs['d_best'][sel].astype(np.int64) + recs['r_best'][sel], lens[pidx[:, 0]].astype(np.int64))
        dr = np.stack([lo, hi], axis=1)
        sa = shared_memory.SharedMemory(create=True, size=arena.nbytes)
        try:
            np.ndarray((arena.nbytes,), np.uint8, buffer=sa.buf)[:] = arena
            n_checked, cells, kernels, dev_ms = 0, 0, set(), 0.0
            for start, stop, b in aligned_batches(arena, offs, lens, pidx, dr, 4, flags=W.PW_FLAG_PROFILE, match_score=1,
                                                  mismatch_score=-3, go_score=-5, ge_score=-2):
                kernels.add(b.kernel_name)
                cells += b.cells; dev_ms += b.fill_ms() + b.trace_ms()
                res = b.results()
                b.pack_transcripts(); b.sync()
                buf, off = b.packed()
                checked, bad = verify.check_packed_parallel(pool, sa, arena.nbytes, offs, lens.astype(np.int64), pidx[start:stop], res,
                                                            buf, off, lo[start:stop], hi[start:stop], (1, -3, -5, -2))
                assert bad == [], (start, bad[:5], len(bad))
                n_checked += checked
        finally:
            sa.close(); sa.unlink()
    assert n_checked == len(sel) and n_checked > 1100000
    assert cells > 2.5e11
    assert all(kn.startswith('k_fill16<') and ', 1>' in kn for kn in kernels), kernels       # the packed overlap rule
    print('config 4 full size: %d alignments, %.3g cells, device %.0f ms (%.0f GCUPS); band selection %.0f ms of device time'
          % (n_checked, cells, dev_ms, cells / dev_ms / 1e6, ms))


def test_config5_full_size():
    """Two 1 Mb sequences sharing 50 planted homologies of 2-20 kb at 80-95 % identity: the local-homology scan must find
    segments covering all 50, and the banded global extension of every segment (the experiment's scores: match
    1 / p_min - 1, mismatch -1, gap extend -1, gap open 0 -- dyadic at p_min = 0.8, so the integer kernels take them) must
    re-score to its reported score exactly; with the non-dyadic scores of p_min = 0.7 (f64 kernel) within rounding."""
    from biseqt_amd import synth, verify
    from biseqt_amd.blot import WordBlot
    from biseqt_amd.pipeline import extend_segments
    from biseqt_amd.sequence import Alphabet, Sequence
    n, k, K_min = 1000000, 12, 1000
    rng = np.random.default_rng(5)
    s = rng.integers(0, 4, n).astype(np.uint8)
    t = rng.integers(0, 4, n).astype(np.uint8)
    planted = []
    for q in range(50):
        ln = int(rng.integers(2000, 20000))
        a = int(rng.integers(0, n - ln)); b = (q * (n // 50) + int(rng.integers(0, 1000))) % (n - ln)
        ident = rng.uniform(.8, .95)
        seg = synth.mutate(rng, s[a:a + ln], (1 - ident) * .7, (1 - ident) * .1, .3)
        seg = seg[:min(len(seg), n - b)]
        t[b:b + len(seg)] = seg
        planted.append((a, b, len(seg)))
    A = Alphabet('ACGT')
    S, T = Sequence(A, tuple(s.tolist())), Sequence(A, tuple(t.tolist()))
    wb = WordBlot(S, T, g_max=.1, sensitivity=.99, alphabet=A, wordlen=k)
    try:
        segs = list(wb.similar_segments(K_min, .7))
    finally:
        wb.close()
    hit = 0
    for (a, b, ln) in planted:
        d, mid = a - b, 2 * b + (a - b) + ln
        hit += any(sg['segment'][0][0] <= d <= sg['segment'][0][1] and sg['segment'][1][0] <= mid <= sg['segment'][1][1] for sg in segs)
    assert hit == 50 and 50 <= len(segs) <= 60, (hit, len(segs))
    for p_min, exact in ((.8, True), (.7, False)):
        match = 1. / p_min - 1
        ext = extend_segments(S, T, segs, k, match_score=match, mismatch_score=-1, ge_score=-1, go_score=0)
        assert len(ext) == len(segs)
        # dyadic scores (0.25 / -1 / 0 / -1) are scaled onto the integer kernels; 3/7 is not dyadic: the f64 kernel
        assert ext[0]['kernel'][1] == ('i32' if exact else 'f64'), ext[0]['kernel']
        n_ok = 0
        for r in ext:
            assert r['alignment'] is not None, r['frame']
            (i0, i1), (j0, j1) = r['frame']
            aln = r['alignment']
            sc, ex, ey, ok = verify.rescore(s[i0:i1], t[j0:j1], aln.transcript, aln.origin_start, aln.mutant_start, match, -1, 0, -1)
            assert ok and (ex, ey) == (i1 - i0, j1 - j0), r['frame']           # a global alignment of the two frames
            assert (sc == r['score']) if exact else abs(sc - r['score']) <= 1e-9 * max(1.0, abs(sc)), (sc, r['score'])
            tr = r['truncated']
            ident = tr.transcript.count('M') / float(len(tr.transcript))
            n_ok += ident > 0.7
        assert n_ok >= 50
