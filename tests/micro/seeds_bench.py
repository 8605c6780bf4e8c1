"""Seed enumeration throughput on one MI355X (config-5 shape: two 1 Mb sequences with planted homologies), next
to the CPU oracle (pure-python restatement of the reference's enumeration) on a bounded sample.

    python tests/micro/seeds_bench.py [n] [--cpu]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from biseqt_amd.seeds import _Index                 # noqa: E402
from biseqt_amd.sequence import Alphabet            # noqa: E402
from biseqt_amd import synth                        # noqa: E402


def genomes(n, seed=5):
    rng = np.random.default_rng(seed)
    s = rng.integers(0, 4, n).astype(np.uint8)
    t = rng.integers(0, 4, n).astype(np.uint8)
    # 50 planted homologies of 2-20 kb at 80-95 % identity
    for _ in range(50):
        ln = int(rng.integers(2000, 20000)) * n // 1000000 or 50
        a, b = int(rng.integers(0, n - ln)), int(rng.integers(0, n - ln))
        seg = synth.mutate(rng, s[a:a + ln], (1 - rng.uniform(.8, .95)) * .7, .02, .3)[:ln]
        t[b:b + len(seg)] = seg
    return s, t


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1000000
    A = Alphabet('ACGT')
    s, t = genomes(n)
    for k in (8, 10, 12, 15, 20):
        with _Index(s, t, k, A, self_comp=0) as idx:
            try:
                idx.build()
            except RuntimeError as e:
                print('k=%2d: %s' % (k, e)); continue
            ms = []
            for _ in range(5):
                nrows = idx.build()
                ms.append(idx.build_ms())
            t_ms = float(np.median(ms))
            t0 = time.perf_counter(); c = idx.count(d_band=(-1000, 1000), a_band=(0, n)); t_cnt = (time.perf_counter() - t0) * 1e3
            ab = idx.algorithmic_bytes()
            print('k=%2d: %11d rows  build %8.3f ms  %7.1f Mkmers/s  %8.1f Mrows/s  alg %6.1f GB/s  band count %d in %.2f ms'
                  % (k, nrows, t_ms, 2 * n / t_ms / 1e3, nrows / t_ms / 1e3, ab / t_ms / 1e6, c, t_cnt))
    if '--cpu' in sys.argv:
        from oracle import seeds_oracle as SO
        m = min(n, 100000)
        t0 = time.perf_counter()
        rows, sc = SO.seed_rows(s[:m].tolist(), t[:m].tolist(), 12, 4)
        dt = time.perf_counter() - t0
        print('cpu oracle (python port, 1 core): %d x %d, k=12: %d rows in %.2f s = %.3f Mkmers/s' % (m, m, len(rows), dt, 2 * m / dt / 1e6))


if __name__ == '__main__':
    main()
