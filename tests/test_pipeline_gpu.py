"""Seeds -> segments -> banded extension (BASELINE config 5 shape, scaled to what the oracle solves in seconds):
frames and bands follow experiments/blot_stats.py:438-453, and every frame's alignment equals the C oracle's
(= the compiled reference's) on the same sub-sequences, transcript for transcript."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_local_homology_scan_vs_oracle(oracle):
    from biseqt_amd import synth
    from biseqt_amd.pipeline import local_homology_scan, segment_frame
    from biseqt_amd.sequence import Alphabet, Sequence
    from oracle import blot_oracle as BO
    A = Alphabet('ACGT')
    rng = synth.rng_for(55)
    n = 30000
    s = synth.rand_seqs(rng, 1, n)[0]
    t = synth.rand_seqs(rng, 1, n)[0]
    planted = []
    for q in range(6):
        ln = int(rng.integers(600, 2500))
        a, b = int(rng.integers(0, n - ln)), q * (n // 6) + int(rng.integers(0, 500))
        seg = synth.mutate(rng, s[a:a + ln], .06, .02, .3)
        seg = seg[:min(len(seg), n - b)]
        t[b:b + len(seg)] = seg
        planted.append((a, b, len(seg)))
    S, T = Sequence(A, tuple(s.tolist())), Sequence(A, tuple(t.tolist()))
    K_min, p_min, wordlen = 400, .8, 10
    segments, ext = local_homology_scan(S, T, K_min, p_min, wordlen, g_max=.1, sensitivity=.99)
    exp_segments = BO.similar_segments(s.tolist(), t.tolist(), wordlen, 4, .1, .99, K_min, p_min)
    assert [x['segment'] for x in segments] == [x['segment'] for x in exp_segments]
    assert len(segments) >= len(planted) and len(ext) == len(segments)
    qM = 1. / p_min - 1
    found = 0
    for seg, rec in zip(segments, ext):
        (i0, i1), (j0, j1), rad = segment_frame(seg['segment'], n, len(T), wordlen)
        assert rec['frame'] == ((i0, i1), (j0, j1)) and rec['diag_range'] == (-rad, rad)
        r = oracle.solve(s[i0:i1], t[j0:j1], L=4, mode=1, alntype=0, diag_range=(-rad, rad), match=qM, mismatch=-1., go=0., ge=-1.)
        if r['init_rc'] != 0 or r['opt'][0] == -1:
            assert rec['alignment'] is None
            continue
        assert rec['score'] == r['score']
        assert rec['alignment'].transcript == r['transcript']
        assert (rec['alignment'].origin_start, rec['alignment'].mutant_start) == (r['origin_idx'], r['mutant_idx'])
        tr = rec['truncated']
        if tr is not None:
            assert tr.transcript[0] == 'M' and tr.transcript[-1] == 'M'
            ident = tr.transcript.count('M') / float(len(tr.transcript))
            found += ident > .75
    assert found >= len(planted)
    for (a, b, ln) in planted:                          # every planted homology is inside some segment's diagonal range
        assert any(x['segment'][0][0] <= a - b <= x['segment'][0][1] for x in segments)
