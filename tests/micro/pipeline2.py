"""Does running two batches on two streams hide the traceback latency behind the other batch's fill?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from biseqt_amd import synth, _pwlib as W
from biseqt_amd.batch import BatchAligner
o, m = synth.pair_batch(2, 10000, 2000)
kw = dict(alnmode=1, alntype=1, alphabet_len=4, diag_range=(-200, 200), match_score=1, mismatch_score=-3, go_score=-5, ge_score=-2)
bs = [BatchAligner(list(zip(o, m)), **kw) for _ in range(4)]
ss = [torch.cuda.Stream() for _ in range(4)]
def run(nb, steps):
    for i in range(4):
        bs[i % nb].solve(ss[i % nb].cuda_stream); bs[i % nb].traceback(ss[i % nb].cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        b, s = bs[i % nb], ss[i % nb].cuda_stream
        b.solve(s); b.traceback(s)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
for nb in (1, 2, 3, 4, 2):
    ms = run(nb, 40)
    print('%d batch(es) in flight: %.3f ms/step, %.1f GCUPS' % (nb, ms, bs[0].cells / ms / 1e6))
