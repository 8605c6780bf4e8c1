"""Batched overlap band selection (include/pw_overlap.h through biseqt_amd.overlap) against the oracle's
highest_scoring_overlap_band of every pair (which runs the reference's KD-tree search on the CPU), then the
config-4 flow: bands -> one banded overlap-alignment batch, alignments equal to the C oracle's."""
import itertools

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _reads(rng, genome_len, n_reads, read_len, subst, gap):
    from biseqt_amd import synth
    g = synth.rand_seqs(rng, 1, genome_len)[0]
    reads, starts = [], []
    for _ in range(n_reads):
        st = int(rng.integers(0, genome_len - read_len))
        reads.append(synth.mutate(rng, g[st:st + read_len], subst, gap, gap))
        starts.append(st)
    return reads, starts


@pytest.mark.parametrize('wordlen,g_max', [(8, .2), (10, .15), (6, .2)])
def test_overlap_bands_vs_oracle(wordlen, g_max):
    from biseqt_amd import synth
    from biseqt_amd.overlap import overlap_bands
    from biseqt_amd.sequence import Alphabet
    from oracle import blot_oracle as BO
    A = Alphabet('ACGT')
    rng = synth.rng_for(1000 + wordlen)
    reads, starts = _reads(rng, 6000, 14, 1500, .04, .03)
    reads.append(reads[0].copy())                          # an identical read: p clamps at 1 on many diagonals
    reads.append(synth.rand_seqs(rng, 1, 700)[0])          # an unrelated, shorter read
    reads.append(np.resize(np.array([0, 1], np.uint8), 400))   # low complexity
    pairs = list(itertools.combinations(range(len(reads)), 2))
    stats = {}
    got = overlap_bands(reads, pairs, wordlen, A, g_max, .99, stats=stats)
    assert stats['fallback_pairs'] < len(pairs) // 2
    nz = 0
    for (i, j), g in zip(pairs, got):
        if len(reads[i]) == len(reads[j]) and (reads[i] == reads[j]).all():
            # documented divergence: the reference turns a pair of identical reads into a SELF comparison
            # (seeds.py:33) and drops the main diagonal; the batch scores them as two different sequences
            assert g['d_band'][0] <= 0 <= g['d_band'][1] and g['p'] > .95
            continue
        e = BO.highest_scoring_overlap_band(reads[i].tolist(), reads[j].tolist(), wordlen, 4, g_max, .99)
        assert (g is None) == (e is None), (i, j)
        if e is None:
            continue
        assert g['d_band'] == e['d_band'] and g['len'] == e['len'], (i, j, g, e)
        assert g['p'] == e['p'] and g['score'] == e['score'], (i, j, g, e)
        nz += g['p'] > .8
    assert nz >= 5


def test_config4_flow_bands_then_banded_overlap_alignment(oracle):
    from biseqt_amd import synth
    from biseqt_amd.overlap import overlap_alignments, overlap_bands
    from biseqt_amd.sequence import Alphabet
    A = Alphabet('ACGT')
    rng = synth.rng_for(4)
    reads, starts = _reads(rng, 20000, 24, 5000, .02, .02)
    pairs = list(itertools.combinations(range(len(reads)), 2))
    bands = overlap_bands(reads, pairs, 10, A, .2, .99)
    alns = overlap_alignments(reads, pairs, bands, A, p_min=.8)
    true_overlap = lambda i, j: min(starts[i], starts[j]) + 5000 - max(starts[i], starts[j])
    found = 0
    for (i, j), band, aln in zip(pairs, bands, alns):
        if true_overlap(i, j) > 800:
            assert band is not None and band['p'] > .8, (i, j, true_overlap(i, j), band)
            assert band['d_band'][0] <= starts[j] - starts[i] <= band['d_band'][1]
            assert aln is not None
            r = oracle.solve(reads[i], reads[j], L=4, mode=1, alntype=2, diag_range=aln['diag_range'],
                             match=1, mismatch=-3, go=-5, ge=-2)
            assert aln['score'] == r['score'] and aln['transcript'] == r['transcript']
            assert (aln['origin_start'], aln['mutant_start']) == (r['origin_idx'], r['mutant_idx'])
            assert aln['score'] > 0.3 * true_overlap(i, j)
            found += 1
        elif true_overlap(i, j) < -200:
            assert band is None or band['p'] < .8
    assert found >= 5


@pytest.mark.parametrize('wordlen', [8, 11])
def test_all_pairs_one_index_vs_pair_list_and_oracle(wordlen):
    """pw_overlap_all_pairs (one k-mer index over all reads, self join) must list exactly the pairs that share a seed
    and give each the record of the pair-list path; a sample is compared with the oracle directly."""
    from biseqt_amd import synth
    from biseqt_amd.overlap import overlap_all_pairs, raw_all_pairs, raw_bands
    from biseqt_amd.sequence import Alphabet
    from oracle import blot_oracle as BO, seeds_oracle as SO
    A = Alphabet('ACGT')
    rng = synth.rng_for(77 + wordlen)
    reads, starts = _reads(rng, 9000, 30, 1200, .04, .03)
    reads.append(synth.rand_seqs(rng, 1, 300)[0])
    reads.append(np.zeros(5, np.uint8))                     # shorter than the word for wordlen 8 / 11
    reads.append(reads[3].copy())
    pairs, recs, ms = raw_all_pairs(reads, wordlen, 4, .2, .99)
    all_pairs = list(itertools.combinations(range(len(reads)), 2))
    ref_recs, _ = raw_bands(reads, all_pairs, wordlen, 4, .2, .99)
    with_seeds = [p for p, r in zip(all_pairs, ref_recs) if r['n_seeds'] > 0]
    assert [tuple(p) for p in pairs.tolist()] == with_seeds
    want = np.array([r for r in ref_recs if r['n_seeds'] > 0], dtype=ref_recs.dtype)
    for name in recs.dtype.names:
        if name != 'pad_':
            assert (recs[name] == want[name]).all(), name
    res = overlap_all_pairs(reads, wordlen, A, .2, .99)
    assert list(res.keys()) == with_seeds
    for (a, b) in with_seeds[::7]:
        if len(reads[a]) == len(reads[b]) and (reads[a] == reads[b]).all():
            continue
        e = BO.highest_scoring_overlap_band(reads[a].tolist(), reads[b].tolist(), wordlen, 4, .2, .99)
        g = res[(a, b)]
        assert g['d_band'] == e['d_band'] and g['p'] == e['p'] and g['len'] == e['len'] and g['score'] == e['score']


def test_all_pairs_shards_partition_the_work():
    """world = 3 simulated on one GPU: the three shards are disjoint, cover exactly the unsharded pair list and carry
    the same records (what raw_all_pairs_sharded gathers)."""
    from biseqt_amd import synth
    from biseqt_amd.overlap import raw_all_pairs
    rng = synth.rng_for(123)
    reads, _ = _reads(rng, 8000, 40, 1000, .03, .02)
    pairs, recs, _ = raw_all_pairs(reads, 10, 4, .2, .99)
    got_p, got_r = [], []
    for rank in range(3):
        p, r, _ = raw_all_pairs(reads, 10, 4, .2, .99, rank=rank, world=3)
        assert (p[:, 0] % 3 == rank).all()
        got_p.append(p); got_r.append(r)
    gp, gr = np.concatenate(got_p), np.concatenate(got_r)
    order = np.lexsort((gp[:, 1], gp[:, 0]))
    assert (gp[order] == pairs).all()
    for name in recs.dtype.names:
        if name != 'pad_':
            assert (gr[order][name] == recs[name]).all(), name


def test_all_pairs_random_read_sets():
    """Random small read sets (repeats, duplicates, reads shorter than the word, two-letter reads) through the
    one-index path -- the sparse one-wavefront kernel for pairs with few seeds and the histogram kernel for the
    others -- against the pair-list path record for record."""
    from biseqt_amd import synth
    from biseqt_amd.overlap import raw_all_pairs, raw_bands
    rng = synth.rng_for(314)
    for trial in range(25):
        R = int(rng.integers(2, 22))
        k = int(rng.integers(3, 11))
        g = synth.rand_seqs(rng, 1, 1500)[0]
        reads = []
        for _ in range(R):
            kind = int(rng.integers(0, 6))
            n = int(rng.integers(0, 500))
            if kind == 0:
                reads.append(synth.rand_seqs(rng, 1, n)[0])
            elif kind == 1:
                st = int(rng.integers(0, 1500 - n)) if n < 1500 else 0
                reads.append(synth.mutate(rng, g[st:st + n], .05, .02, .3) if n else g[:0].copy())
            elif kind == 2:
                reads.append(np.resize(rng.integers(0, 4, int(rng.integers(1, 4))).astype(np.uint8), n))
            elif kind == 3 and reads:
                reads.append(reads[int(rng.integers(0, len(reads)))].copy())
            elif kind == 4:
                reads.append(rng.integers(0, 2, n).astype(np.uint8))
            else:
                reads.append(g[int(rng.integers(0, 700)):][:n].copy())
        g_max, sens = float(rng.choice([.1, .2, .3])), float(rng.choice([.9, .99]))
        pairs, recs, _ = raw_all_pairs(reads, k, 4, g_max, sens)
        allp = [(a, b) for a in range(R) for b in range(a + 1, R)]
        ref, _ = raw_bands(reads, allp, k, 4, g_max, sens)
        keep = [q for q, r in enumerate(ref) if r['n_seeds'] > 0]
        assert [tuple(p) for p in pairs.tolist()] == [allp[q] for q in keep], trial
        for name in recs.dtype.names:
            if name != 'pad_':
                assert (recs[name] == ref[name][keep]).all(), (trial, name)


def test_all_pairs_job_every_alignment_rescored():
    """The config-4 flow on a job of 2 000 reads of 5 kb (25x coverage of a 400 kb genome): all pairs through one index,
    the pairs with p >= 0.8 through banded overlap alignment in chunked batches that refer to one uploaded read arena,
    and EVERY alignment checked by the size-independent properties: re-scoring the transcript reproduces the score, every
    M / S agrees with the letters, the path ends in the reported cell, starts on the table edge and ends on the last row
    or column (an overlap alignment), and stays inside its band.  Recall against the known read positions."""
    from biseqt_amd import synth, verify, _pwlib as W
    from biseqt_amd.batch import BatchAligner, pack_reads
    from biseqt_amd.overlap import raw_all_pairs
    R, read_len, cov, k = 2000, 5000, 25, 16
    rng = synth.rng_for(404)
    G = R * read_len // cov
    g = synth.rand_seqs(rng, 1, G)[0]
    starts = rng.integers(0, G - read_len, R)
    reads = [synth.mutate(rng, g[s:s + read_len], .05, .025, .025) for s in starts]
    pairs, recs, _ = raw_all_pairs(reads, k, 4, .2, .9)
    w = recs['w_best']
    p = np.where(w > 0, np.exp(np.log(np.maximum(w, 1e-300)) / k), 0.0)
    sel = np.flatnonzero(p >= .8)
    ov = np.minimum(starts[pairs[:, 0]], starts[pairs[:, 1]]) + read_len - np.maximum(starts[pairs[:, 0]], starts[pairs[:, 1]])
    order = np.argsort(starts); ss = starts[order]
    true_total = int(sum(np.searchsorted(ss, ss[i] + read_len - 500, 'left') - i - 1 for i in range(R)))
    found = int((ov[sel] > 500).sum())
    assert found >= 0.99 * true_total and int((ov[sel] <= 0).sum()) <= 0.01 * len(sel) + 2, (found, true_total, len(sel))
    arena, offs, lens = pack_reads(reads)
    pidx = pairs[sel].astype(np.int64)
    lo = np.maximum(recs['d_best'][sel].astype(np.int64) - recs['r_best'][sel], -lens[pidx[:, 1]].astype(np.int64))
    hi = np.minimum(recs['d_best'][sel].astype(np.int64) + recs['r_best'][sel], lens[pidx[:, 0]].astype(np.int64))
    dr = np.stack([lo, hi], axis=1)
    cells = (hi - lo + 1) * np.minimum(lens[pidx[:, 0]], lens[pidx[:, 1]]).astype(np.int64)
    start, n_checked, kernels = 0, 0, set()
    while start < len(sel):
        stop = start + max(1, int(np.searchsorted(np.cumsum(cells[start:]), 2 * 10 ** 10, 'right')))
        with BatchAligner.from_arena(arena, offs, lens, pidx[start:stop], dr[start:stop], alnmode=W.BANDED_MODE,
                                     alntype=W.B_OVERLAP, alphabet_len=4, match_score=1, mismatch_score=-3, go_score=-5,
                                     ge_score=-2) as b:
            kernels.add(b.kernel_name)
            res = b.run()
            txs = b.transcripts(res)
        o_l = [reads[a] for a in pidx[start:stop, 0]]
        m_l = [reads[c] for c in pidx[start:stop, 1]]
        assert (res['opt_i'] >= 0).all() and ((res['status'] & 15) == W.PW_ST_TRACED).all()
        assert verify.check_batch(o_l, m_l, res, txs, 1, -3, -5, -2, banded=True, dmins=lo[start:stop].tolist()) == []
        for q in range(stop - start):
            X, Y = len(o_l[q]), len(m_l[q])
            x0, y0 = int(res['origin_idx'][q]), int(res['mutant_idx'][q])
            ex, ey = verify.end_cell_xy(res['opt_i'][q], res['opt_j'][q], True, lo[start + q])
            assert (x0 == 0 or y0 == 0) and (ex == X or ey == Y), q
            ops = np.frombuffer(txs[q].encode(), np.uint8)
            d = x0 - y0 + np.cumsum((ops == 68).astype(np.int64) - (ops == 73).astype(np.int64))
            assert lo[start + q] <= min(d.min(), x0 - y0) and max(d.max(), x0 - y0) <= hi[start + q], q
        n_checked += stop - start
        start = stop
    assert n_checked == len(sel) and n_checked > 30000
    assert all('k_fill16' in kn for kn in kernels), kernels


def test_shared_device_arena_and_pipelined_batches_equal_private_copies():
    """One device copy of the reads for all batches of a job (pw_arena_upload / PW_FLAG_SHARED_ARENA) and the pipelined
    batch loop (the next batch is planned while one runs) return what batches with their own arena copy return; a batch
    created for a shared arena refuses to run without one and refuses an upload."""
    from biseqt_amd import _pwlib as W
    from biseqt_amd import synth
    from biseqt_amd.batch import BatchAligner, DeviceArena, pack_reads
    from biseqt_amd.overlap import aligned_batches
    rng = synth.rng_for(4242)
    genome = synth.rand_seqs(rng, 1, 30000)[0]
    reads = []
    for _ in range(120):
        a = int(rng.integers(0, 30000 - 1500))
        reads.append(synth.mutate(rng, genome[a:a + int(rng.integers(600, 1500))], 0.03, 0.01, 0.5))
    arena, offs, lens = pack_reads(reads)
    pidx = np.array([(i, j) for i in range(120) for j in range(i + 1, 120) if (i * 7 + j) % 9 == 0], np.int64)
    dr = np.stack([np.maximum(-lens[pidx[:, 1]].astype(np.int64), -60), np.minimum(lens[pidx[:, 0]].astype(np.int64), 60)], axis=1)
    kw = dict(alnmode=W.BANDED_MODE, alntype=W.B_OVERLAP, alphabet_len=4, match_score=1, mismatch_score=-3, go_score=-5, ge_score=-2)
    with BatchAligner.from_arena(arena, offs, lens, pidx, dr, **kw) as b:
        ref = b.run().copy()
        ref_tx = b.transcripts(ref)
    got, got_tx, nb = [], [], 0
    for start, stop, b in aligned_batches(arena, offs, lens, pidx, dr, 4, max_cells=3 * 10 ** 6, match_score=1, mismatch_score=-3,
                                          go_score=-5, ge_score=-2):
        res = b.results()
        got.append(res.copy()); got_tx.extend(b.transcripts(res)); nb += 1
    assert nb >= 3
    got = np.concatenate(got)
    assert (got == ref).all() and got_tx == ref_tx
    with DeviceArena(arena) as dev:
        b = BatchAligner.from_arena(arena, offs, lens, pidx[:50], dr[:50], device_arena=dev, **kw)
        try:
            with pytest.raises(RuntimeError):
                b.upload()
            assert (b.run() == ref[:50]).all()
        finally:
            b.close()
