"""Config-4 shape at single-GPU test scale: synthetic 5 kb reads of a random genome at ~25x coverage, all pairs of
the first R reads: overlap band selection for every pair in one device pass, then banded overlap alignment of the
pairs with p >= 0.8 in one batch.

    python tests/micro/overlap_bench.py [n_reads] [wordlen]
"""
import itertools
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from biseqt_amd import synth                                     # noqa: E402
from biseqt_amd.overlap import overlap_alignments, overlap_bands, raw_bands   # noqa: E402
from biseqt_amd.sequence import Alphabet                         # noqa: E402


def main():
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    rng = synth.rng_for(4)
    read_len, cov = 5000, 25
    G = R * read_len // cov
    g = synth.rand_seqs(rng, 1, G)[0]
    starts = rng.integers(0, G - read_len, R)
    reads = [synth.mutate(rng, g[s:s + read_len], .05, .025, .025) for s in starts]
    pairs = list(itertools.combinations(range(R), 2))
    A = Alphabet('ACGT')
    raw_bands(reads[:8], pairs[:4], k, 4, .2, .9)                # warm-up (module load)
    t0 = time.perf_counter()
    recs, ms = raw_bands(reads, pairs, k, 4, .2, .9)
    t1 = time.perf_counter()
    stats = {}
    bands = overlap_bands(reads, pairs, k, A, .2, .9, stats=stats)
    t2 = time.perf_counter()
    alns = overlap_alignments(reads, pairs, bands, A, p_min=.8)
    t3 = time.perf_counter()
    kmers = sum(len(reads[i]) + len(reads[j]) for i, j in pairs)
    tp = fp = fn = 0
    for (i, j), b in zip(pairs, bands):
        ov = min(starts[i], starts[j]) + read_len - max(starts[i], starts[j])
        pos = b is not None and b['p'] >= .8
        tp += pos and ov > 500; fp += pos and ov <= 0; fn += (not pos) and ov > 500
    na = sum(a is not None for a in alns)
    print('%d reads, %d pairs, k=%d: band selection device %.1f ms (%.2f G k-mers/s, %.2f M pairs/s), wall %.2f s; '
          'host scoring %.2f s (%d tie fallbacks); %d seeds total' % (R, len(pairs), k, ms, kmers / ms / 1e6, len(pairs) / ms / 1e3,
                                                                      t1 - t0, t2 - t1, stats['fallback_pairs'], int(recs['n_seeds'].sum())))
    print('classifier p >= 0.8: %d true overlaps (> 500 bases) found, %d missed, %d false; banded overlap alignment of %d pairs: %.2f s wall'
          % (tp, fn, fp, na, t3 - t2))


if __name__ == '__main__':
    main()
