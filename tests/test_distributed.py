"""The N > 1 path on CPU: world_size-2 `gloo` run of the round-robin shard + result gather
(biseqt_amd/distributed.py).  The per-rank "device results" are produced by the oracle here (no GPU
in this container); what is under test is the partitioning, padding, gather and re-ordering."""
import os
import socket

import numpy as np
import pytest

from biseqt_amd import synth
from biseqt_amd.batch import RESULT_DTYPE
from biseqt_amd.distributed import shard_indices


def test_shard_indices_partition_and_balance():
    n, world = 103, 8
    parts = [shard_indices(n, r, world) for r in range(world)]
    assert sorted(np.concatenate(parts).tolist()) == list(range(n))
    assert [p.tolist() for p in parts][3][:3] == [3, 11, 19]                 # pair p -> rank p mod world
    w = np.random.default_rng(0).integers(1, 1000, n)
    parts = [shard_indices(n, r, world, weights=w) for r in range(world)]
    assert sorted(np.concatenate(parts).tolist()) == list(range(n))
    loads = [w[p].sum() for p in parts]
    assert max(loads) - min(loads) <= w.max()


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_pairs, tmpdir):
    import torch
    import torch.distributed as dist
    from oracle import oracle as O
    from biseqt_amd.distributed import gather_bytes, gather_records
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    origins, mutants = synth.pair_batch(11, n_pairs, 60)
    mine = shard_indices(n_pairs, rank, world)
    rec = np.zeros(len(mine), RESULT_DTYPE)
    txs = []
    for k, p in enumerate(mine):
        r = O.solve(origins[p], mutants[p], L=4, mode=1, alntype=O.B_LOCAL, diag_range=(-10, 10),
                    match=1, mismatch=-3, go=-5, ge=-2)
        rec[k] = (r['score'], r['opt'][0], r['opt'][1], r['origin_idx'], r['mutant_idx'],
                  len(r['transcript']), 1)
        txs.append(r['transcript'])
    full = gather_records(rec, n_pairs, rank, world)
    blob = torch.from_numpy(np.frombuffer(''.join(txs).encode(), dtype=np.uint8).copy())
    blobs = gather_bytes(blob, rank, world)
    if rank == 0:
        np.save(os.path.join(tmpdir, 'full.npy'), full)
        with open(os.path.join(tmpdir, 'tx.txt'), 'w') as f:
            f.write('\n'.join(bytes(b.numpy()).decode() for b in blobs))
    else:
        assert full is None and blobs is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_gather(tmp_path, oracle):
    import torch.multiprocessing as mp
    n_pairs, world = 9, 2          # odd on purpose: ranks hold 5 and 4 pairs (ragged gather)
    mp.spawn(_worker, args=(world, _free_port(), n_pairs, str(tmp_path)), nprocs=world, join=True)
    full = np.load(os.path.join(str(tmp_path), 'full.npy'))
    origins, mutants = synth.pair_batch(11, n_pairs, 60)
    txcat = {0: '', 1: ''}
    for p in range(n_pairs):
        r = oracle.solve(origins[p], mutants[p], L=4, mode=1, alntype=oracle.B_LOCAL, diag_range=(-10, 10),
                         match=1, mismatch=-3, go=-5, ge=-2)
        assert full['score'][p] == r['score'] and (full['opt_i'][p], full['opt_j'][p]) == r['opt'], p
        assert full['tx_len'][p] == len(r['transcript'])
        txcat[p % world] += r['transcript']
    got = open(os.path.join(str(tmp_path), 'tx.txt')).read().split('\n')
    assert got == [txcat[0], txcat[1]]


def _worker_struct(rank, world, port, n_total, outdir):
    import torch.distributed as dist
    from biseqt_amd.distributed import gather_struct, shard_indices
    from biseqt_amd.overlap import BAND_DTYPE
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    mine = shard_indices(n_total, rank, world)
    recs = np.zeros(len(mine), BAND_DTYPE)
    recs['d_best'] = mine * 3 - 7                      # a value that identifies the global pair index
    recs['w_best'] = mine / 8.0
    recs['n_seeds'] = mine + 1
    full = gather_struct(recs, n_total, rank, world)
    if rank == 0:
        np.save(os.path.join(outdir, 'bands.npy'), full)
    else:
        assert full is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_gather_of_overlap_band_records(tmp_path):
    """The config-4 shard: pair q on rank q mod 2, 64-byte band records gathered to rank 0 in pair order."""
    import torch.multiprocessing as mp
    n_total, world = 11, 2
    mp.spawn(_worker_struct, args=(world, _free_port(), n_total, str(tmp_path)), nprocs=world, join=True)
    full = np.load(os.path.join(str(tmp_path), 'bands.npy'))
    q = np.arange(n_total)
    assert (full['d_best'] == q * 3 - 7).all() and (full['w_best'] == q / 8.0).all() and (full['n_seeds'] == q + 1).all()


def _worker_ragged(rank, world, port, outdir):
    import torch
    import torch.distributed as dist
    from biseqt_amd import distributed as D
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    n = 1000 + 37 * rank                                     # ragged: every rank a different byte count
    mine = torch.from_numpy(((np.arange(n) * (rank + 3)) % 251).astype(np.uint8))
    sizes = D.exchange_sizes(n, rank, world)
    assert sizes == [1000 + 37 * r for r in range(world)]
    recv = [torch.zeros(s, dtype=torch.uint8) for s in sizes] if rank == 0 else None
    for step in range(3):                                    # the same buffers every step, as bench.py does
        D.gather_ragged_wait(D.gather_ragged_start(mine, recv, rank, world))
    if rank == 0:
        for r in range(1, world):
            assert (recv[r].numpy() == ((np.arange(sizes[r]) * (r + 3)) % 251).astype(np.uint8)).all()
        open(os.path.join(outdir, 'ok'), 'w').write('ok')
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_ragged_transcript_gather(tmp_path):
    """bench.py's per-step transcript gather: sizes agreed once, then one grouped send / receive per step."""
    import torch.multiprocessing as mp
    mp.spawn(_worker_ragged, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(os.path.join(str(tmp_path), 'ok'))


def test_bench_self_launch_refuses_without_enough_devices():
    """`python bench.py --gpus N` starts its N ranks itself; with fewer than N devices it says so and exits 2
    (no assertion error, no GPU call)."""
    import subprocess
    import sys
    import torch
    have = torch.cuda.device_count()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    want = max(have + 1, 2)
    p = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', str(want)], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=600)
    assert p.returncode == 2
    assert 'needs %d visible devices' % want in p.stderr


def _worker_delayed(rank, world, port, outdir):
    import torch
    import torch.distributed as dist
    from biseqt_amd import distributed as D
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    cap = 5000
    caps = [cap] * world
    nfl = 2                                                   # two slots in flight, as bench.py keeps two batches
    slots = [D.DelayedRaggedGather(rank, world, caps) for _ in range(nfl)]
    bufs = [torch.zeros(cap, dtype=torch.uint8) for _ in range(nfl)]
    def size_of(step, r):
        return 0 if (step == 2 and r == 1) else 700 + 131 * step + 17 * r      # data dependent, one empty payload

    def content(step, r, n):
        return ((np.arange(n) * (r + 3) + step) % 251).astype(np.uint8)

    for step in range(5):
        j = step % nfl
        out = slots[j].collect()                              # the payload of this slot's previous step, if any
        if out is not None and rank == 0:                     # (checked before the slot's buffer is written again: rank 0's
            for r in range(world):                            #  own part is a view of it) -- it belongs to step - nfl
                exp = content(step - nfl, r, size_of(step - nfl, r))
                assert out[r].numel() == len(exp) and (out[r].numpy() == exp).all(), (step, r)
        n = size_of(step, rank)
        bufs[j][:n] = torch.from_numpy(content(step, rank, n))
        mine = torch.tensor([n], dtype=torch.int64)
        totals = [torch.zeros(1, dtype=torch.int64) for _ in range(world)] if rank == 0 else None
        dist.gather(mine, totals, dst=0)                      # the step's fixed-size traffic carries the byte counts
        slots[j].post(bufs[j], (lambda n=n: n), torch.cat(totals) if rank == 0 else None)
    for j in range(nfl):                                      # drain
        out = slots[j].collect()
        if rank == 0:
            s = [st for st in (3, 4) if st % nfl == j][0]
            for r in range(world):
                exp = content(s, r, size_of(s, r))
                assert out[r].numel() == len(exp) and (out[r].numpy() == exp).all(), (s, r)
    assert all(sl.collect() is None for sl in slots)
    if rank == 0:
        open(os.path.join(outdir, 'ok'), 'w').write('ok')
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_delayed_gather_of_packed_transcripts(tmp_path):
    """bench.py's gather of the PACKED transcripts: byte counts are data dependent and known on the device only, so they
    ride with the step's fixed-size gather and the bytes follow one use of the slot later (DelayedRaggedGather)."""
    import torch.multiprocessing as mp
    mp.spawn(_worker_delayed, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(os.path.join(str(tmp_path), 'ok'))
