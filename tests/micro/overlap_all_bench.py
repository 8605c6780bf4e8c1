"""Config 4 through ONE index over all reads: R synthetic 5 kb reads at 25x coverage, every pair of reads.

    python tests/micro/overlap_all_bench.py [n_reads] [wordlen]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from biseqt_amd import synth                            # noqa: E402
from biseqt_amd.overlap import raw_all_pairs            # noqa: E402
from biseqt_amd.batch import BatchAligner             # noqa: E402
from biseqt_amd import _pwlib as W                     # noqa: E402


def main():
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    rng = synth.rng_for(4)
    read_len, cov = 5000, 25
    G = R * read_len // cov
    g = synth.rand_seqs(rng, 1, G)[0]
    starts = rng.integers(0, G - read_len, R)
    t0 = time.perf_counter()
    reads = [synth.mutate(rng, g[s:s + read_len], .05, .025, .025) for s in starts]
    t1 = time.perf_counter()
    raw_all_pairs(reads[:50], k, 4, .2, .9)
    t2 = time.perf_counter()
    pairs, recs, ms = raw_all_pairs(reads, k, 4, .2, .9, max_pairs=min(R * (R - 1) // 2, 1 << 23))
    t3 = time.perf_counter()
    w = recs['w_best']
    p = np.where(w > 0, np.exp(np.log(np.maximum(w, 1e-300)) / k), 0.0)
    pos = p >= .8
    ov = np.minimum(starts[pairs[:, 0]], starts[pairs[:, 1]]) + read_len - np.maximum(starts[pairs[:, 0]], starts[pairs[:, 1]])
    # true overlaps among ALL pairs (sorted starts sweep)
    order = np.argsort(starts); ss = starts[order]
    true_total = int(sum(np.searchsorted(ss, ss[i] + read_len - 500, 'left') - i - 1 for i in range(R)))
    print('%d reads (%.1f Mb of reads, genome %d), %d pairs, k=%d: one index, device %.1f ms (wall %.2f s; reads generated in %.1f s)'
          % (R, R * read_len / 1e6, G, R * (R - 1) // 2, k, ms, t3 - t2, t1 - t0))
    print('  candidate pairs sharing a seed: %d (%d seeds); p >= 0.8: %d pairs, of which overlapping > 500 bases: %d, not overlapping: %d; '
          'true overlaps > 500 bases in the read set: %d' % (len(pairs), int(recs['n_seeds'].sum()), int(pos.sum()),
                                                            int((pos & (ov > 500)).sum()), int((pos & (ov <= 0)).sum()), true_total))
    print('  %.2f G pairs/s of the all-pairs space, %.2f M candidate pairs/s' % (R * (R - 1) / 2 / ms / 1e6, len(pairs) / ms / 1e3))
    # banded overlap alignment (B_OVERLAP, 1/-3/-5/-2) of the positive pairs (the first NA of them; NA < 0: all), the
    # reads uploaded once and referred to by the pairs, in batches of at most 2e10 cells
    from biseqt_amd.batch import pack_reads
    NA = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
    sel = np.flatnonzero(pos)
    if NA >= 0:
        sel = sel[:NA]
    t4 = time.perf_counter()
    arena, offs, lens = pack_reads(reads)
    pidx = pairs[sel].astype(np.int64)
    lo = np.maximum(recs['d_best'][sel].astype(np.int64) - recs['r_best'][sel], -lens[pidx[:, 1]].astype(np.int64))
    hi = np.minimum(recs['d_best'][sel].astype(np.int64) + recs['r_best'][sel], lens[pidx[:, 0]].astype(np.int64))
    dr = np.stack([lo, hi], axis=1)
    cells = (hi - lo + 1) * np.minimum(lens[pidx[:, 0]], lens[pidx[:, 1]]).astype(np.int64)
    if len(sel):
        nd = hi - lo + 1
        print('  bands of the pairs to align: %d .. %d diagonals, mean %.1f, median %d, 90 %% below %d' %
              (nd.min(), nd.max(), nd.mean(), np.median(nd), np.percentile(nd, 90)))
    t5 = time.perf_counter()
    dev_ms, tot_cells, nb, score_sum, kname = 0.0, 0, 0, 0.0, ''
    from biseqt_amd.overlap import aligned_batches
    for start, stop, b in aligned_batches(arena, offs, lens, pidx, dr, 4, max_cells=int(os.environ.get('OV_MAX_CELLS', 2 * 10 ** 10)),
                                          flags=W.PW_FLAG_PROFILE, match_score=1, mismatch_score=-3, go_score=-5, ge_score=-2):
        res = b.results()
        dev_ms += b.fill_ms() + b.trace_ms(); tot_cells += b.cells; nb += 1; score_sum += float(res['score'].sum())
        kname = b.kernel_name
    t6 = time.perf_counter()
    if len(sel):
        print('  banded overlap alignment of %d pairs in %d batches: %.3g cells, device %.1f ms = %.0f GCUPS (%s); wall %.2f s '
              '(+ %.2f s packing the reads); mean score %.0f' % (len(sel), nb, tot_cells, dev_ms, tot_cells / dev_ms / 1e6, kname,
                                                            t6 - t5, t5 - t4, score_sum / max(len(sel), 1)))


if __name__ == '__main__':
    main()
