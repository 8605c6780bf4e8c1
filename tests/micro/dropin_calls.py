import os, sys, time, ctypes as C
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from biseqt_amd import synth, _pwlib as W
from oracle import ref_driver as R
lib = R.load(W.PWLIB_SO)
rng = synth.rng_for(1)
a, b = synth.rand_seqs(rng, 2, 1000)
for sc in (dict(match=1., mismatch=0., go=0., ge=0.), dict(match=1., mismatch=-3., go=-5., ge=-2.)):
    P = R.Problem(a.tolist(), b.tolist(), mode=0, alntype=0, L=4, **sc)
    for rep in range(3):
        T = P.table
        t = [time.perf_counter()]
        lib.dptable_init(C.byref(T)); t.append(time.perf_counter())
        opt = lib.dptable_solve(C.byref(T)); t.append(time.perf_counter())
        aln = lib.dptable_traceback(C.byref(T), opt); t.append(time.perf_counter())
        lib.dptable_free(C.byref(T)); t.append(time.perf_counter())
    print(sc, 'init %.2f solve %.2f traceback %.2f free %.2f ms' % tuple((t[i+1]-t[i])*1e3 for i in range(4)))
