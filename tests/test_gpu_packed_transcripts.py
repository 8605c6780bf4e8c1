"""The device-side compaction of the transcripts (`pw_batch_pack_transcripts`: exclusive scan of tx_len + one copy kernel),
i.e. the bytes the multi-GPU gather moves (bench.py --gather packed): the packed buffer must be the transcripts of
`pw_batch_transcripts` back to back in pair order, the offsets their running sum -- ragged batches, pairs without an
alignment, pairs dptable_init rejects, one long strip-pipeline transcript, and after re-solving the same batch."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _check(b, res, txs):
    from biseqt_amd.batch import BatchAligner
    b.pack_transcripts()
    buf, off = b.packed()
    lens = np.maximum(res['tx_len'], 0).astype(np.int64)
    assert off[0] == 0 and (np.diff(off.astype(np.int64)) == lens).all()
    assert int(off[-1]) == lens.sum() == buf.size
    assert BatchAligner.transcripts_from_packed(buf, off) == txs
    assert bytes(buf).decode('ascii') == ''.join(t or '' for t in txs)


def test_packed_transcripts_equal_the_slots(oracle):
    from biseqt_amd import synth
    from biseqt_amd.batch import BatchAligner
    rng = synth.rng_for(3601)
    pairs = []
    for k in range(700):
        n = int(rng.integers(0, 500)) if k % 9 else int(rng.integers(0, 3))
        o = synth.rand_seqs(rng, 1, n)[0]
        if k % 4 == 0:                                      # nothing in common: often no local alignment at all
            m = ((o + 2) % 4)[: max(0, n - int(rng.integers(0, 5)))].astype(np.uint8) if n else synth.rand_seqs(rng, 1, 3)[0]
        else:
            m = synth.mutate(rng, o, 0.06, 0.03, 0.4)
        pairs.append((o, m))
    # banded local: empty transcripts occur; banded global with a band that is infeasible for some pairs: init_rc = -1
    for kw in (dict(alnmode=1, alntype=1, diag_range=(-25, 25), match_score=1, mismatch_score=-3, go_score=-5, ge_score=-2),
               dict(alnmode=1, alntype=0, diag_range=(-6, 6), match_score=1, mismatch_score=-1, go_score=0, ge_score=-1),
               dict(alnmode=0, alntype=1, match_score=2, mismatch_score=-3, go_score=-4, ge_score=-1)):
        with BatchAligner(pairs, alphabet_len=4, check_band=False, **kw) as b:
            res = b.run()
            txs = b.transcripts(res)
            _check(b, res, txs)
            if kw['alnmode'] == 1 and kw['alntype'] == 0:
                assert any(b.init_rc(k) != 0 for k in range(len(pairs)))
            assert any(t is None for t in txs) and any(t for t in txs)
            res2 = b.run()                                   # the same batch again: same bytes
            _check(b, res2, b.transcripts(res2))
    k = 5
    r = oracle.solve(pairs[k][0], pairs[k][1], L=4, mode=0, alntype=1, match=2, mismatch=-3, go=-4, ge=-1)
    assert (txs[k] or None) == r['transcript']


def test_packed_transcripts_with_a_strip_pipeline_pair():
    """One wide standard-mode pair (strip layout, a 12 000-op transcript) in the batch."""
    from biseqt_amd import _pwlib as W
    from biseqt_amd import synth
    from biseqt_amd.batch import BatchAligner
    rng = synth.rng_for(3602)
    pairs = []
    for n in (6000, 300, 1200):
        o = synth.rand_seqs(rng, 1, n)[0]
        pairs.append((o, synth.mutate(rng, o, 0.08, 0.04, 0.3)))
    with BatchAligner(pairs, alnmode=0, alntype=0, alphabet_len=4, match_score=1, mismatch_score=-3, go_score=-5, ge_score=-2,
                      flags=W.PW_FLAG_FORCE_STRIP) as b:
        res = b.run()
        txs = b.transcripts(res)
        assert len(txs[0]) >= 6000
        _check(b, res, txs)


def test_packed_total_arrives_asynchronously():
    from biseqt_amd import synth
    from biseqt_amd.batch import BatchAligner, PinnedArray
    origins, mutants = synth.pair_batch(5, 500, 400)
    with BatchAligner(list(zip(origins, mutants)), alnmode=1, alntype=1, alphabet_len=4, diag_range=(-40, 40), match_score=1,
                      mismatch_score=-3, go_score=-5, ge_score=-2) as b:
        pin = PinnedArray(8, np.uint64)
        b.solve(); b.traceback(); b.pack_transcripts(); b.packed_total_async(pin); b.sync()
        res = b.results()
        assert int(pin.array[0]) == int(np.maximum(res['tx_len'], 0).sum()) > 0
        pin.close()
