#!/usr/bin/env python
"""Headline benchmark: GCUPS of banded local alignment (BASELINE.json config 2) on N MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path (fill + end-cell search + traceback, pw_batch_solve +
pw_batch_traceback) over one batch of 10 000 synthetic 2 kb x ~2 kb pairs, band radius 200, B_LOCAL,
scores match 1 / mismatch -3 / gap open -5 / gap extend -2, inputs resident in HBM.  With N > 1 every
rank holds its own batch of the same shape (pairs dealt round-robin from an N-times larger job: weak
scaling) and each step ends with the gather of the 32-byte result records to rank 0 over RCCL.  Two batches of
that shape are kept in flight per GPU, each on its own HIP stream, consecutive steps alternating between them
(--inflight): the traceback is a latency-bound walk, and the other batch's fill hides it.

Prints ONE JSON line (rank 0).  `roofline` prices the fill kernel (the dominant kernel) with the
algorithmic bytes of SURVEY.md 8d -- 0.5 B per cell (4-bit tie mask) + X + Y + 32 B per pair + the
transcript bytes -- over its mean duration measured with HIP events on the launch stream inside the
library.  `cpu_baseline` times the reference pwlib itself (oracle/_ref, compiled from the reference
sources) -- or the oracle restatement if that file is absent -- on the host cores, on a bounded sample
of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

PAIRS, LENGTH, RADIUS = 10000, 2000, 200
SCORES = dict(match=1., mismatch=-3., go=-5., ge=-2.)
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


# ---------------------------------------------------------------------------------------------------
# CPU baseline (rank 0, N == 1): the reference library on the host cores, bounded sample
# ---------------------------------------------------------------------------------------------------
def _cpu_worker(args):
    kind, seed, first, count, budget_s = args
    sys.path.insert(0, ROOT)
    from biseqt_amd import synth
    origins, mutants = synth.pair_batch(seed, first + count, LENGTH)
    cells = 0
    t0 = time.time()
    done = 0
    if kind == 'reference':
        from oracle import ref_driver as R
        lib = R.load()
        devnull = os.open(os.devnull, os.O_WRONLY)
        os.dup2(devnull, 1)           # the reference prints band messages to stdout
        for k in range(first, first + count):
            P = R.Problem(origins[k].tolist(), mutants[k].tolist(), mode=R.BANDED_MODE, alntype=R.B_LOCAL,
                          diag_range=(-RADIUS, RADIUS), L=4, **SCORES)
            R.run(lib, P)              # init + solve + traceback + free, as pw.py drives it
            cells += synth.banded_cells(len(origins[k]), len(mutants[k]), -RADIUS, RADIUS)
            done += 1
            if time.time() - t0 > budget_s:
                break
    else:
        from oracle import oracle as O
        for k in range(first, first + count):
            O.solve(origins[k], mutants[k], L=4, mode=1, alntype=O.B_LOCAL, diag_range=(-RADIUS, RADIUS), **SCORES)
            cells += synth.banded_cells(len(origins[k]), len(mutants[k]), -RADIUS, RADIUS)
            done += 1
            if time.time() - t0 > budget_s:
                break
    return cells, done, time.time() - t0


def cpu_baseline(budget_s=12.0):
    import multiprocessing as mp
    kind = 'reference' if os.path.exists(os.path.join(ROOT, 'oracle', '_ref', 'pwlib_ref.so')) else 'port'
    if kind == 'port':
        from oracle import oracle as O
        O.lib()
    cores = min(os.cpu_count() or 1, 16)
    per = 64
    ctx = mp.get_context('spawn')     # never fork a process that may touch the GPU
    t0 = time.time()
    with ctx.Pool(cores) as pool:
        out = pool.map(_cpu_worker, [(kind, 2, c * per, per, budget_s) for c in range(cores)])
    wall = time.time() - t0
    cells = sum(o[0] for o in out)
    npairs = sum(o[1] for o in out)
    busy = max(o[2] for o in out)
    return dict(value=cells / busy / 1e9, unit='GCUPS', cores=cores, kind=kind,
                sample='%d pairs of the cfg2 batch (2 kb x ~2 kb, band radius 200, B_LOCAL), %d single-threaded '
                       'processes, init+solve+traceback+free per pair, %.1f s busy / %.1f s wall'
                       % (npairs, cores, busy, wall))


def pmc_traffic(kernel, pairs):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/pmc_traffic.json: WRITE_SIZE + 2 x FETCH_SIZE, KiB -> bytes, per MI355X_MICROARCH.md's gfx950
    correction), valid only for the kernel and batch size they were collected on; else None."""
    path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    try:
        with open(path) as f:
            rec = json.load(f)
        if rec.get('kernel') == kernel and rec.get('pairs') == pairs:
            return rec['hbm_bytes_per_launch']
    except (OSError, ValueError, KeyError):
        pass
    return None


# ---------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--pairs', type=int, default=PAIRS, help='pairs per GPU (default: the BASELINE config)')
    ap.add_argument('--inflight', type=int, default=2,
                    help='batches in flight per GPU, each on its own HIP stream (consecutive steps alternate); 2 hides the '
                         'latency-bound traceback of one batch behind the fill of the next (measured: 4.53 -> 3.88 ms/step)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--force-dist', action='store_true', help='run the RCCL gather path even with one rank (self-test)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    assert world == args.gpus, 'launch with torch.distributed.run --nproc-per-node == --gpus'

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()           # before anything touches the GPU

    import torch
    import torch.distributed as dist
    from biseqt_amd import _pwlib as W
    from biseqt_amd import synth
    from biseqt_amd.batch import BatchAligner, RESULT_DTYPE

    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29513')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    # rank r owns pairs r, r + world, ... of a job of world * pairs pairs (round-robin shard).  `inflight` batches
    # of that shape (different synthetic pairs) are resident; step i runs batch i % inflight on stream i % inflight.
    n_local = args.pairs
    nfl = max(1, args.inflight)
    batches, streams = [], []
    for j in range(nfl):
        origins, mutants = synth.pair_batch(2 + 1000 * rank + 100 * j, n_local, LENGTH)
        batches.append(BatchAligner(list(zip(origins, mutants)), alnmode=W.BANDED_MODE, alntype=W.B_LOCAL, alphabet_len=4,
                                    diag_range=(-RADIUS, RADIUS), match_score=SCORES['match'],
                                    mismatch_score=SCORES['mismatch'], go_score=SCORES['go'], ge_score=SCORES['ge'],
                                    device=local_rank, flags=W.PW_FLAG_PROFILE))
        streams.append(torch.cuda.Stream(device=dev))
    batch = batches[0]
    cells = batch.cells
    res_devs = [torch.as_tensor(b.results_device(), device=dev) for b in batches] if use_dist else None
    gathered = [[torch.empty(32 * n_local, dtype=torch.uint8, device=dev) for _ in range(world)] for _ in range(nfl)] \
        if (use_dist and rank == 0) else None

    def step(i):
        j = i % nfl
        with torch.cuda.stream(streams[j]):
            s = streams[j].cuda_stream
            batches[j].solve(s)
            batches[j].traceback(s)
            if use_dist:
                dist.gather(res_devs[j], gathered[j] if rank == 0 else None, dst=0)
        return batches[j].cells

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    fill_ms, trace_ms = [], []
    done_cells = 0
    t0 = time.perf_counter()
    for i in range(args.steps):
        done_cells += step(i)
    fence()
    elapsed = time.perf_counter() - t0
    # fill-kernel duration of the LAST timed step of each batch (HIP events on its stream): with two batches in flight
    # the two fills co-run, so each takes about twice as long as alone while two complete per that time
    overlapped_ms = [b.fill_ms() for b in batches[:min(nfl, args.steps)]]
    # per-kernel durations (HIP events recorded by the library on the launch stream): sample a few extra steps of
    # ONE batch alone, outside the timed region, so that the kernel time is not stretched by the other batch
    stream = streams[0].cuda_stream
    for _ in range(min(5, max(1, args.steps))):
        batch.solve(stream)
        batch.traceback(stream)
        batch.sync(stream)
        fill_ms.append(batch.fill_ms())
        trace_ms.append(batch.trace_ms())
    t = torch.tensor([elapsed, float(done_cells)], dtype=torch.float64, device=dev)
    if use_dist:
        tm = t[:1].clone(); dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        tc = t[1:].clone(); dist.all_reduce(tc, op=dist.ReduceOp.SUM)
        elapsed, total_done = float(tm.item()), float(tc.item())
    else:
        total_done = float(done_cells)

    res = batch.results()
    tx_bytes = int(res['tx_len'].sum())
    alg_bytes = batch.algorithmic_bytes + tx_bytes
    fill = float(np.mean(fill_ms))
    # parity spot check inside the bench: re-score a few transcripts (cheap, size-independent)
    ok = bool((res['status'] & 1).all() and (res['opt_i'] >= 0).all())

    if rank == 0:
        value = total_done / elapsed / 1e9          # cells of every step of every rank / max-over-ranks time
        achieved = alg_bytes / (fill * 1e-3) / 1e9
        line = {
            'metric': 'GCUPS (DP cell updates/s) banded local align',
            'value': round(value, 3), 'unit': 'GCUPS', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 4),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            # the arithmetic the dominant kernel computes in: packed 16-bit lanes for k_fill16, else i32 / f64
            'dtype': 'i16' if 'k_fill16' in batch.kernel_name else ('i32' if batch.score_dtype == 'i32' else 'f64'),
            'data': 'synthetic',
            'config': {'workload': 'BASELINE configs[1]: %d pairs/GPU, %d x ~%d, band radius %d, B_LOCAL, '
                                   'match 1 / mismatch -3 / go -5 / ge -2, fill + end-cell search + traceback%s; '
                                   '%d batches in flight per GPU on separate HIP streams'
                                   % (n_local, LENGTH, LENGTH, RADIUS,
                                      ' + RCCL gather of result records' if world > 1 else '', nfl),
                       'pairs_per_gpu': n_local, 'cells_per_gpu': int(cells), 'batches_in_flight': nfl, 'parallelism': 'pairs round-robin x%d' % world},
            'roofline': {'bound': 'hbm', 'achieved': round(achieved, 2), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': round(achieved / HBM_PEAK_GBS, 5), 'traffic': pmc_traffic(batch.kernel_name, n_local),
                         'kernel': batch.kernel_name,
                         'kernel_ms': round(fill, 4), 'kernel_ms_note': 'one batch running alone (5 extra steps after the '
                         'timed region); in the timed region %d fills co-run, each lasting %s ms' % (nfl, '/'.join('%.2f' % v for v in overlapped_ms)),
                         'algorithmic_bytes_per_launch': int(alg_bytes),
                         'kernel_gcups': round(cells / (fill * 1e-3) / 1e9, 2),
                         'traceback_kernel_ms': round(float(np.mean(trace_ms)), 4)},
            'results_ok': ok,
        }
        if cpu is not None:
            line['cpu_baseline'] = cpu
        print(json.dumps(line))
    if use_dist and rank == 0:
        # the gathered records of rank 0 must be this rank's device records, bit for bit
        got = gathered[0][0].cpu().numpy().view(RESULT_DTYPE)
        assert (got == res).all(), 'gathered records differ from the local results'
    for b in batches:
        b.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
