"""Soak of the four drop-in calls: thousands of init / solve / traceback / free cycles must not leak host or device
memory and must keep a steady latency.    python tests/micro/soak_dropin.py [iterations]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch                                        # noqa: E402  (device memory statistics only)
from biseqt_amd import synth, _pwlib as W           # noqa: E402
from oracle import ref_driver as R                  # noqa: E402  (the ctypes driver of the pwlib ABI)


def rss_mb():
    with open('/proc/self/statm') as f:
        return int(f.read().split()[1]) * os.sysconf('SC_PAGE_SIZE') / 2.0 ** 20


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    lib = R.load(W.PWLIB_SO)
    o, m = synth.pair_batch(2, 8, 2000)
    rng = synth.rng_for(9)
    a, b = synth.rand_seqs(rng, 2, 300)
    probs = [R.Problem(o[k].tolist(), m[k].tolist(), mode=1, alntype=1, diag_range=(-200, 200), L=4, match=1., mismatch=-3., go=-5., ge=-2.) for k in range(8)]
    probs.append(R.Problem(a.tolist(), b.tolist(), mode=0, alntype=0, L=4, match=1., mismatch=-1., go=-2., ge=-1.))   # STD: full table
    devnull = os.open(os.devnull, os.O_WRONLY); saved = os.dup(1); os.dup2(devnull, 1)
    first = [R.run(lib, P) for P in probs]
    marks = []
    for it in range(n):
        t0 = time.perf_counter()
        out = R.run(lib, probs[it % len(probs)])
        dt = time.perf_counter() - t0
        assert out['transcript'] == first[it % len(probs)]['transcript'] and out['score'] == first[it % len(probs)]['score']
        if it % (n // 6) == 0 or it == n - 1:
            free, total = torch.cuda.mem_get_info()
            marks.append((it, rss_mb(), (total - free) / 2.0 ** 20, dt * 1e3))
    os.dup2(saved, 1)
    for it, rss, dev, ms in marks:
        print('iteration %5d: host RSS %8.1f MiB, device memory in use %8.1f MiB, this call %.2f ms' % (it, rss, dev, ms))
    grow_host = marks[-1][1] - marks[1][1]
    grow_dev = marks[-1][2] - marks[1][2]
    print('growth after warm-up: host %.1f MiB, device %.1f MiB' % (grow_host, grow_dev))
    assert grow_host < 64 and grow_dev < 64, 'memory grows with the number of calls'
    print('OK')


if __name__ == '__main__':
    main()
