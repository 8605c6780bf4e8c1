#!/usr/bin/env python
"""Headline benchmark: GCUPS of banded local alignment (BASELINE.json config 2) on N MI355X.

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no torch.distributed environment the script starts N child ranks itself (``python -m
torch.distributed.run`` as a subprocess, before anything touches a GPU); under ``torch.distributed.run`` it is one
rank.  It needs N visible devices and says so if there are fewer.

One "step" = one pass of the hot path (fill + end-cell search + traceback: pw_batch_solve + pw_batch_traceback) over
one batch of 10 000 synthetic 2 kb x ~2 kb pairs, band radius 200, B_LOCAL, scores match 1 / mismatch -3 / gap open -5 /
gap extend -2, inputs resident in HBM.  With N > 1 every rank holds its own batch of the same shape (pairs dealt
round-robin from an N-times larger job: weak scaling) and each step ends with the gather of the 32-byte result records
AND of the transcript slots to rank 0 over RCCL.  Two batches are kept in flight per GPU, each on its own HIP stream
(--inflight): the traceback is a latency-bound walk, and the other batch's fill hides it.

Prints ONE JSON line (rank 0):
  value / ms_per_step     the timed K steps (all ranks, max over ranks)
  roofline                the fill kernel (the dominant kernel) priced with the algorithmic bytes of SURVEY.md 8d --
                          0.5 B per cell (4-bit tie mask) + X + Y + 32 B per pair + the transcript bytes -- over its
                          mean duration, HIP events recorded by the library on the launch stream while one batch runs
                          alone (the `serial` leg: this is the figure a rocprofv3 --kernel-trace of
                          `bench.py --inflight 1` shows); `frac_timed_region` prices the same bytes over the timed
                          region's wall time per step instead
  serial                  the same steps with one batch in flight
  e2e_with_h2d_d2h        steps that also upload the sequences from pinned host memory and bring records and
                          transcripts back; never `value`
  variants                the 32-bit kernel (the like-for-like width of the reference's integer results), the f64 kernel
                          (the reference's own arithmetic) and the linear-gap rerun (go 0) on the same pairs; config 1
                          through the four drop-in calls
  roofline.valu_*         VALU instructions of the dominant kernel from the committed rocprofv3 PMC passes
                          (profiles/pmc_kernel.json), reported only while the built kernel's code fingerprint equals the
                          one the counters were collected on (biseqt_amd/pwlib/kernel_hashes.json); null otherwise
  check                   what was verified about the batches the clock ran on
  cpu_baseline            the reference pwlib itself (oracle/_ref, the reference's own -g flags) -- or the oracle
                          restatement -- on the host cores, bounded sample of the same batch; its outputs are the
                          `check.vs_reference` comparison.  cpu_baseline_O2: the same sources at -O2

--min-seconds S (default 1): the timed region runs max(K, enough steps for S seconds) steps -- decided from the warm-up
before the clock starts, `steps` reports what ran, `steps_requested` the K given; --min-seconds 0 times exactly K.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

PAIRS, LENGTH, RADIUS = 10000, 2000, 200
SCORES = dict(match=1., mismatch=-3., go=-5., ge=-2.)
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
N_SIMD, CLOCK_GHZ = 1024, 2.4   # MI355X_MICROARCH.md: 256 CUs x 4 SIMDs, 2.4 GHz peak engine clock
CPU_SAMPLE_PER_CORE = 64


def batch_seed(rank, j):
    return 2 + 1000 * rank + 100 * j


# ---------------------------------------------------------------------------------------------------
# CPU baseline (rank 0, N == 1): the reference library on the host cores, bounded sample of batch 0
# ---------------------------------------------------------------------------------------------------
def _cpu_worker(args):
    kind, first, origins, mutants, budget_s = args[:5]
    so_path = args[5] if len(args) > 5 else None
    sys.path.insert(0, ROOT)
    from biseqt_amd import synth
    cells = 0
    out = []
    t0 = time.time()
    if kind == 'reference':
        from oracle import ref_driver as R
        lib = R.load(so_path) if so_path else R.load()
        devnull = os.open(os.devnull, os.O_WRONLY)
        os.dup2(devnull, 1)           # the reference prints band messages to stdout
        for k in range(len(origins)):
            P = R.Problem(origins[k].tolist(), mutants[k].tolist(), mode=R.BANDED_MODE, alntype=R.B_LOCAL,
                          diag_range=(-RADIUS, RADIUS), L=4, **SCORES)
            r = R.run(lib, P)          # init + solve + traceback + free, as pw.py drives it
            out.append((first + k, r['score'], tuple(r['opt']), r['transcript'], r['origin_idx'], r['mutant_idx']))
            cells += synth.banded_cells(len(origins[k]), len(mutants[k]), -RADIUS, RADIUS)
            if time.time() - t0 > budget_s:
                break
    else:
        from oracle import oracle as O
        for k in range(len(origins)):
            r = O.solve(origins[k], mutants[k], L=4, mode=1, alntype=O.B_LOCAL, diag_range=(-RADIUS, RADIUS), **SCORES)
            out.append((first + k, r['score'], tuple(r['opt']), r['transcript'], r['origin_idx'], r['mutant_idx']))
            cells += synth.banded_cells(len(origins[k]), len(mutants[k]), -RADIUS, RADIUS)
            if time.time() - t0 > budget_s:
                break
    return cells, out, time.time() - t0


def cpu_baseline(origins, mutants, budget_s=12.0, so_name='pwlib_ref.so', flags='-g, no -O: the reference Makefile\'s own flags'):
    """Times the reference on the first pairs of the batch the GPU will run; returns (json object, reference answers)."""
    import multiprocessing as mp
    so_path = os.path.join(ROOT, 'oracle', '_ref', so_name)
    kind = 'reference' if os.path.exists(so_path) else 'port'
    if kind == 'port':
        from oracle import oracle as O
        O.lib()
    cores = min(os.cpu_count() or 1, 16)
    per = min(CPU_SAMPLE_PER_CORE, max(1, len(origins) // cores))
    ctx = mp.get_context('spawn')     # never fork a process that may touch the GPU
    t0 = time.time()
    with ctx.Pool(cores) as pool:
        out = pool.map(_cpu_worker, [(kind, c * per, origins[c * per:(c + 1) * per], mutants[c * per:(c + 1) * per], budget_s, so_path)
                                     for c in range(cores)])
    wall = time.time() - t0
    cells = sum(o[0] for o in out)
    answers = [a for o in out for a in o[1]]
    busy = max(o[2] for o in out)
    obj = dict(value=cells / busy / 1e9, unit='GCUPS', cores=cores, kind=kind,
               sample='%d pairs of the timed cfg2 batch (2 kb x ~2 kb, band radius 200, B_LOCAL), %d single-threaded '
                      'processes, init+solve+traceback+free per pair, %.1f s busy / %.1f s wall'
                      % (len(answers), cores, busy, wall))
    if kind == 'reference':
        obj['build'] = 'oracle/_ref/%s (gcc %s)' % (so_name, flags)
    return obj, answers


def pmc_counters(kernel, pairs):
    """Counters of the dominant kernel from the committed rocprofv3 PMC passes (profiles/pmc_kernel.json), per launch:
    HBM bytes (WRITE_SIZE + 2 x FETCH_SIZE, KiB -> bytes, per MI355X_MICROARCH.md's gfx950 correction) and VALU
    instructions.  They describe ONE code object: the record carries the fingerprint of the kernel's instruction stream at
    collection time, and it is used only while biseqt_amd/pwlib/kernel_hashes.json (written by the build) shows the same
    fingerprint for the kernel that just ran, on the same batch size.  Returns (record or None, reason)."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'pmc_kernel.json')) as f:
            rec = json.load(f)
        with open(os.path.join(ROOT, 'biseqt_amd', 'pwlib', 'kernel_hashes.json')) as f:
            built = json.load(f)
    except (OSError, ValueError) as e:
        return None, 'no counter record or no kernel fingerprints (%s)' % e
    if rec.get('kernel') != kernel or rec.get('pairs') != pairs:
        return None, 'counters were collected on %s, %s pairs' % (rec.get('kernel'), rec.get('pairs'))
    if built.get(rec.get('symbol')) != rec.get('code_sha256'):
        return None, 'stale: the kernel was rebuilt with different code since the counters were collected'
    return rec, 'profiles/pmc_kernel.json, code fingerprint %s' % rec['code_sha256'][:12]


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(args, argv):
    """--gpus N > 1 without a torch.distributed environment: start the N ranks as children.  Nothing in this process
    has touched a GPU yet (counting devices does not initialise one on this image)."""
    import torch
    have = torch.cuda.device_count()
    if have < args.gpus:
        sys.stderr.write('bench.py: --gpus %d needs %d visible devices, this machine shows %d\n' % (args.gpus, args.gpus, have))
        return 2
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd)


# ---------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--pairs', type=int, default=PAIRS, help='pairs per GPU (default: the BASELINE config)')
    ap.add_argument('--inflight', type=int, default=2,
                    help='batches in flight per GPU, each on its own HIP stream (consecutive steps alternate); 2 hides the '
                         'latency-bound traceback of one batch behind the fill of the next')
    ap.add_argument('--min-seconds', type=float, default=1.0,
                    help='lower bound of the timed region: more than --steps steps are timed if needed (0: exactly --steps)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='skip the serial / e2e / variant legs (profiling runs)')
    ap.add_argument('--force-dist', action='store_true', help='run the RCCL gather path even with one rank (self-test)')
    ap.add_argument('--gather', choices=('packed', 'slots'), default='packed',
                    help='what the transcript gather moves: the ops compacted on the device (pw_batch_pack_transcripts, about '
                         'half the bytes; they follow one use of the batch later, when their size is known) or the X + Y + 1 '
                         'byte slots as they are')
    args = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args, sys.argv[1:]))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        sys.exit('bench.py: --gpus %d but the launcher started %d ranks' % (args.gpus, world))

    from biseqt_amd import synth, verify
    n_local = args.pairs
    nfl = max(1, args.inflight)
    seqs = [synth.pair_batch(batch_seed(rank, j), n_local, LENGTH) for j in range(nfl)]

    cpu, cpu_o2, ref_answers = None, None, []
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu, ref_answers = cpu_baseline(seqs[0][0], seqs[0][1])     # before anything touches the GPU
        if cpu['kind'] == 'reference' and os.path.exists(os.path.join(ROOT, 'oracle', '_ref', 'pwlib_ref_O2.so')):
            cpu_o2, o2_answers = cpu_baseline(seqs[0][0], seqs[0][1], budget_s=8.0, so_name='pwlib_ref_O2.so', flags='-O2')
            same = {a[0]: a for a in ref_answers}
            cpu_o2['answers_equal_the_g_build'] = all(same.get(a[0], a) == a for a in o2_answers)

    import torch
    import torch.distributed as dist
    from biseqt_amd import _pwlib as W
    from biseqt_amd.batch import BatchAligner, PinnedArray, RESULT_DTYPE
    from biseqt_amd import distributed as D

    if torch.cuda.device_count() <= local_rank:
        sys.exit('bench.py: rank %d needs device %d, %d visible' % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    use_dist = world > 1 or args.force_dist
    # stdout carries exactly one JSON line: libraries that print banners there (RCCL's version block at communicator
    # set-up) are sent to stderr for the rest of the run, the line goes to the original descriptor at the end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29513')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    akw = dict(alnmode=W.BANDED_MODE, alntype=W.B_LOCAL, alphabet_len=4, diag_range=(-RADIUS, RADIUS),
               match_score=SCORES['match'], mismatch_score=SCORES['mismatch'], go_score=SCORES['go'],
               ge_score=SCORES['ge'], device=local_rank)
    # rank r owns pairs r, r + world, ... of a job of world * pairs pairs (round-robin shard).  `inflight` batches
    # of that shape (different synthetic pairs) are resident; step i runs batch i % inflight on stream i % inflight.
    batches = [BatchAligner(list(zip(*seqs[j])), flags=W.PW_FLAG_PROFILE, **akw) for j in range(nfl)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(nfl)]
    batch = batches[0]
    cells = batch.cells
    res_devs = tx_devs = gathered = gathered_tx = None
    packed_gather = use_dist and args.gather == 'packed'
    if use_dist:
        res_devs = [torch.as_tensor(b.results_device(), device=dev) for b in batches]
        tx_sizes = [D.exchange_sizes(b.transcripts_bytes, rank, world, device=dev) for b in batches]
        if rank == 0:
            gathered = [[torch.empty(32 * n_local, dtype=torch.uint8, device=dev) for _ in range(world)] for _ in range(nfl)]
        if packed_gather:
            for b in batches:                                  # (allocates the packed buffer and the offsets)
                b.pack_transcripts(None); b.sync(None)
            packed_devs = [torch.as_tensor(b.packed_device()[0], device=dev) for b in batches]
            # the byte count of a step = the last entry of the offsets the scan writes: 8 bytes on the device ...
            total_devs = [torch.as_tensor(b.packed_device()[1], device=dev)[8 * n_local:].view(torch.int64) for b in batches]
            total_pins = [PinnedArray(8, np.uint64) for _ in batches]            # ... and their copy on the host
            gathered_totals = [[torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)] for _ in range(nfl)] if rank == 0 else None
            delayed = [D.DelayedRaggedGather(rank, world, tx_sizes[j], device=dev) for j in range(nfl)]
            moved = []
        else:
            tx_devs = [torch.as_tensor(b.transcripts_device(), device=dev) for b in batches]
            if rank == 0:      # (rank 0 keeps its own slots where they are: no receive buffer for itself)
                gathered_tx = [[None] + [torch.empty(tx_sizes[j][r], dtype=torch.uint8, device=dev) for r in range(1, world)] for j in range(nfl)]

    def step(i):
        j = i % nfl
        with torch.cuda.stream(streams[j]):
            s = streams[j].cuda_stream
            if packed_gather:
                # the packed transcripts of this batch's PREVIOUS step leave now: their size has long reached the host,
                # and the buffers are about to be written again
                out = delayed[j].collect()
                if out is not None:
                    moved.append(sum(int(t.numel()) for t in out[1:]))
            batches[j].solve(s)
            batches[j].traceback(s)
            if use_dist:
                # the single gather of scores and tracebacks (north star): 32-byte records, then the transcripts
                dist.gather(res_devs[j], gathered[j] if rank == 0 else None, dst=0)
                if packed_gather:
                    batches[j].pack_transcripts(s)
                    batches[j].packed_total_async(total_pins[j], s)
                    dist.gather(total_devs[j], gathered_totals[j] if rank == 0 else None, dst=0)
                    ev = torch.cuda.Event(); ev.record(streams[j])

                    def own_total(ev=ev, pin=total_pins[j]):
                        ev.synchronize()
                        return int(pin.array[0])
                    delayed[j].post(packed_devs[j], own_total, torch.cat(gathered_totals[j]) if rank == 0 else None)
                else:
                    D.gather_ragged_wait(D.gather_ragged_start(tx_devs[j], gathered_tx[j] if rank == 0 else None, rank, world))
        return batches[j].cells

    def drain():
        if packed_gather:
            for j in range(nfl):
                with torch.cuda.stream(streams[j]):
                    out = delayed[j].collect()
                    if out is not None:
                        moved.append(sum(int(t.numel()) for t in out[1:]))

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    # how many steps the timed region needs to last --min-seconds: measured on a few untimed steps, agreed over the ranks
    steps_run = args.steps
    if args.min_seconds > 0 and args.steps > 0:
        np_ = max(2 * nfl, 4)
        tp = time.perf_counter()
        for i in range(np_):
            step(i)
        fence()
        est = (time.perf_counter() - tp) / np_
        need = torch.tensor([int(np.ceil(args.min_seconds / max(est, 1e-6)))], dtype=torch.int64, device=dev)
        if use_dist:
            dist.all_reduce(need, op=dist.ReduceOp.MAX)
        steps_run = max(args.steps, min(int(need.item()), 100000))
    drain()
    fence()
    done_cells = 0
    t0 = time.perf_counter()
    for i in range(steps_run):
        done_cells += step(i)
    drain()                 # the last steps' transcripts arrive inside the timed region
    fence()
    elapsed = time.perf_counter() - t0
    # fill-kernel duration of the LAST timed step of each batch: with two batches in flight the fills co-run
    overlapped_ms = [b.fill_ms() for b in batches[:min(nfl, steps_run)]]

    t = torch.tensor([elapsed, float(done_cells)], dtype=torch.float64, device=dev)
    if use_dist:
        tm = t[:1].clone(); dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        tc = t[1:].clone(); dist.all_reduce(tc, op=dist.ReduceOp.SUM)
        elapsed, total_done = float(tm.item()), float(tc.item())
    else:
        total_done = float(done_cells)

    # ---- what the clock ran on, checked: every transcript of every timed batch re-scored on the host ------------
    check = {}
    results, transcripts = [], []
    n_bad = 0
    for j, b in enumerate(batches):
        res = b.results()
        txs = b.transcripts(res)
        results.append(res); transcripts.append(txs)
        traced = bool(((res['status'] & 1) == 1).all() and (res['opt_i'] >= 0).all())
        bad = verify.check_batch(seqs[j][0], seqs[j][1], res, txs, SCORES['match'], SCORES['mismatch'], SCORES['go'],
                                 SCORES['ge'], banded=True, dmins=[-RADIUS] * n_local)
        n_bad += len(bad) + (0 if traced else 1)
    check['rescored_pairs'] = n_local * nfl
    check['rescore_failures'] = n_bad
    n_ref_bad = 0
    for (k, score, opt, tx, oi, mi) in ref_answers:             # rank 0, N == 1: the reference's own answers
        r = results[0][k]
        if not (r['score'] == score and (int(r['opt_i']), int(r['opt_j'])) == tuple(opt) and transcripts[0][k] == tx
                and (int(r['origin_idx']), int(r['mutant_idx'])) == (oi, mi)):
            n_ref_bad += 1
    check['vs_reference_pairs'] = len(ref_answers)
    check['vs_reference_mismatches'] = n_ref_bad
    if use_dist:
        bad_t = torch.tensor([n_bad], dtype=torch.int64, device=dev)
        dist.all_reduce(bad_t, op=dist.ReduceOp.SUM)
        check['rescore_failures'] = int(bad_t.item())
        check['rescored_pairs'] = n_local * nfl * world
        if rank == 0:
            # what arrived at the root: rank 0's own records bit for bit, and the last rank's records + transcripts,
            # regenerated here from its seed and re-scored
            last_j = (steps_run - 1) % nfl if steps_run else 0
            got0 = gathered[last_j][0].cpu().numpy().view(RESULT_DTYPE)
            gather_ok = bool((got0 == results[last_j]).all())
            r_last = world - 1
            if r_last > 0:
                o_l, m_l = synth.pair_batch(batch_seed(r_last, last_j), n_local, LENGTH)
                rec_l = gathered[last_j][r_last].cpu().numpy().view(RESULT_DTYPE)
                if packed_gather:
                    # the ops back to back in pair order: the offsets are the running sum of the records' lengths
                    blob = delayed[last_j].last[r_last].cpu().numpy()
                    offs = np.concatenate([[0], np.cumsum(np.maximum(rec_l['tx_len'], 0))]).astype(np.int64)
                    gather_ok = gather_ok and int(offs[-1]) == blob.size
                    txs_l = BatchAligner.transcripts_from_packed(blob, offs)
                else:
                    slots = gathered_tx[last_j][r_last].cpu().numpy()
                    # the slot layout is a function of the lengths alone (pw_batch_tx_slot): cap = X + Y + 1, 16-byte aligned
                    caps = np.array([len(o) + len(m) + 1 for o, m in zip(o_l, m_l)], np.int64)
                    offs = np.concatenate([[0], np.cumsum((caps + 15) // 16 * 16)[:-1]])
                    txs_l = [slots[offs[k] + caps[k] - rec_l['tx_len'][k]: offs[k] + caps[k]].tobytes().decode('ascii')
                             for k in range(n_local)]
                gather_ok = gather_ok and verify.check_batch(o_l, m_l, rec_l, txs_l, SCORES['match'], SCORES['mismatch'],
                                                             SCORES['go'], SCORES['ge'], banded=True,
                                                             dmins=[-RADIUS] * n_local) == []
            if packed_gather:
                # rank 0's own packed buffer of the last step against its own slots
                own = delayed[last_j].last[0].cpu().numpy()
                offs0 = np.concatenate([[0], np.cumsum(np.maximum(results[last_j]['tx_len'], 0))]).astype(np.int64)
                gather_ok = gather_ok and BatchAligner.transcripts_from_packed(own, offs0) == transcripts[last_j]
            check['gathered_records_and_transcripts_ok'] = gather_ok
            check['transcript_gather'] = args.gather
            if packed_gather and moved:
                check['transcript_bytes_received_per_step'] = int(np.mean(moved))
                check['transcript_slot_bytes_per_step'] = int(sum(tx_sizes[0][1:]))
    ok = check['rescore_failures'] == 0 and n_ref_bad == 0 and check.get('gathered_records_and_transcripts_ok', True)

    # ---- extra legs (outside the timed region) ----------------------------------------------------------------------
    extras = {}
    stream0 = streams[0].cuda_stream
    nx = max(3, min(10, args.steps))
    # (1) serial: one batch in flight; the library's HIP events on the launch stream give the fill kernel alone
    fill_ms, trace_ms = [], []
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(nx):
        batch.solve(stream0); batch.traceback(stream0); batch.sync(stream0)
        fill_ms.append(batch.fill_ms()); trace_ms.append(batch.trace_ms())
    serial_ms = (time.perf_counter() - t1) / nx * 1e3
    fill = float(np.mean(fill_ms))
    tx_bytes = int(results[0]['tx_len'].sum())
    alg_bytes = batch.algorithmic_bytes + tx_bytes
    extras['serial'] = {'batches_in_flight': 1, 'ms_per_step': round(serial_ms, 4), 'gcups': round(cells / serial_ms / 1e6, 2)}
    variants = {}
    if not args.no_extras and rank == 0:
        # (2) end to end with the host boundary: H2D of the sequence arena, fill, traceback, D2H of records + transcripts
        # (three batches in flight here: a batch's copies then overlap the kernels of TWO others -- measured 4.8 ms per
        #  step with two in flight, 4.2 ms with three; the resident loop above gains nothing from a third)
        e_nfl = max(nfl, 3)
        e_batches, e_streams, e_results = list(batches), list(streams), list(results)
        for j in range(nfl, e_nfl):
            sq = synth.pair_batch(batch_seed(rank, j), n_local, LENGTH)
            e_batches.append(BatchAligner(list(zip(*sq)), flags=W.PW_FLAG_PROFILE, **akw))
            e_streams.append(torch.cuda.Stream(device=dev))
            e_results.append(e_batches[-1].run().copy())
        pins = []
        for b in e_batches:
            pa = PinnedArray(b.arena.nbytes); pa.array[:] = b.arena
            pins.append((pa, PinnedArray(32 * n_local), PinnedArray(b.transcripts_bytes)))

        def e2e_step(i):
            j = i % e_nfl
            s = e_streams[j].cuda_stream
            pa, pr, pt = pins[j]
            e_batches[j].upload_async(pa, s)
            e_batches[j].solve(s); e_batches[j].traceback(s)
            e_batches[j].results_async(pr, s); e_batches[j].transcripts_async(pt, s)

        for i in range(e_nfl):
            e2e_step(i)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(nx):
            e2e_step(i)
        torch.cuda.synchronize()
        e2e_ms = (time.perf_counter() - t1) / nx * 1e3
        # the loop above ran e_nfl + nx steps: the last one used batch (e_nfl + nx - 1) % e_nfl
        last = (e_nfl + nx - 1) % e_nfl
        back = pins[last][1].array.view(RESULT_DTYPE)
        e_cells = float(np.mean([b.cells for b in e_batches]))
        extras['e2e_with_h2d_d2h'] = {
            'ms_per_step': round(e2e_ms, 4), 'gcups': round(e_cells / e2e_ms / 1e6, 2), 'batches_in_flight': e_nfl,
            'h2d_bytes': int(batch.arena.nbytes), 'd2h_bytes': int(32 * n_local + batch.transcripts_bytes),
            'host_memory': 'pinned (pw_host_alloc)', 'records_equal_resident_run': bool((back == e_results[last]).all())}
        ok = ok and extras['e2e_with_h2d_d2h']['records_equal_resident_run']
        for pa, pr, pt in pins:
            pa.close(); pr.close(); pt.close()
        for b in e_batches[nfl:]:
            b.close()
        # (3) the same pairs through the 32-bit kernel, and with linear gaps (go 0), each checked by re-scoring
        for name, flags, go in (('int32_kernel', W.PW_FLAG_NO_PACKED16 | W.PW_FLAG_PROFILE, SCORES['go']),
                                ('f64_kernel', W.PW_FLAG_FORCE_F64 | W.PW_FLAG_PROFILE, SCORES['go']),
                                ('linear_gap_go0', W.PW_FLAG_PROFILE, 0.0)):
            kw2 = dict(akw); kw2['go_score'] = go
            with BatchAligner(list(zip(*seqs[0])), flags=flags, **kw2) as b2:
                fm, wall = [], []
                for _ in range(4):
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    b2.solve(stream0); b2.traceback(stream0); b2.sync(stream0)
                    wall.append((time.perf_counter() - t1) * 1e3); fm.append(b2.fill_ms())
                r2 = b2.results(); x2 = b2.transcripts(r2)
                f2 = float(np.mean(fm[1:])); ab2 = b2.algorithmic_bytes + int(r2['tx_len'].sum())
                bad2 = verify.check_batch(seqs[0][0], seqs[0][1], r2, x2, SCORES['match'], SCORES['mismatch'], go,
                                          SCORES['ge'], banded=True, dmins=[-RADIUS] * n_local)
                same = bool((r2 == results[0]).all()) if go == SCORES['go'] else None
                variants[name] = {'kernel': b2.kernel_name, 'kernel_ms': round(f2, 4), 'kernel_gcups': round(cells / f2 / 1e6, 2),
                                  'hbm_frac': round(ab2 / (f2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                  'ms_per_step_serial': round(float(np.mean(wall[1:])), 4),
                                  'gcups_serial': round(cells / float(np.mean(wall[1:])) / 1e6, 2),
                                  'rescore_failures': len(bad2)}
                if same is not None:
                    variants[name]['records_equal_packed16_run'] = same
                    ok = ok and same
                ok = ok and not bad2
        # (4) BASELINE config 1 through the four drop-in calls (1 kb x 1 kb GLOBAL, default scores and 1/-3/-5/-2)
        from biseqt_amd.pw import Aligner
        from biseqt_amd.sequence import Alphabet
        A = Alphabet('ACGT')
        rng = synth.rng_for(1)
        s1 = A.parse(''.join('ACGT'[c] for c in synth.rand_seqs(rng, 1, 1000)[0]))
        s2 = A.parse(''.join('ACGT'[c] for c in synth.rand_seqs(rng, 1, 1000)[0]))
        cfg1 = {}
        for name, kw1 in (('default_scores', {}), ('scores_1_-3_-5_-2', dict(match_score=1, mismatch_score=-3, go_score=-5, ge_score=-2))):
            ms = []
            for _ in range(3):
                t1 = time.perf_counter()
                with Aligner(s1, s2, alnmode=W.STD_MODE, alntype=W.GLOBAL, **kw1) as al:
                    sc = al.solve()
                    aln = al.traceback()
                ms.append((time.perf_counter() - t1) * 1e3)
                rescored = al.calculate_score(aln)
            cfg1[name] = {'ms_init_solve_traceback_free': round(min(ms), 3), 'score': sc, 'rescored': rescored}
            ok = ok and rescored == sc
        variants['config1_dropin_1kb_global'] = cfg1

    if rank == 0:
        value = total_done / elapsed / 1e9          # cells of every step of every rank / max-over-ranks time
        ms_step = elapsed / steps_run * 1e3
        achieved = alg_bytes / (fill * 1e-3) / 1e9
        pmc, pmc_src = pmc_counters(batch.kernel_name, n_local)
        valu = pmc.get('SQ_INSTS_VALU') if pmc else None
        line = {
            'metric': 'GCUPS (DP cell updates/s) banded local align',
            'value': round(value, 3), 'unit': 'GCUPS', 'n_gpus': world, 'steps': steps_run, 'steps_requested': args.steps,
            'timed_region_s': round(elapsed, 4),
            'warmup': args.warmup, 'ms_per_step': round(ms_step, 4),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            # the arithmetic the dominant kernel computes in: packed 16-bit lanes for k_fill16, else i32 / f64
            'dtype': 'i16' if 'k_fill16' in batch.kernel_name else ('i32' if batch.score_dtype == 'i32' else 'f64'),
            'data': 'synthetic',
            'config': {'workload': 'BASELINE configs[1]: %d pairs/GPU, %d x ~%d, band radius %d, B_LOCAL, '
                                   'match 1 / mismatch -3 / go -5 / ge -2, fill + end-cell search + traceback%s; '
                                   '%d batches in flight per GPU on separate HIP streams'
                                   % (n_local, LENGTH, LENGTH, RADIUS,
                                      ' + RCCL gather of result records and transcripts to rank 0' if world > 1 else '', nfl),
                       'pairs_per_gpu': n_local, 'cells_per_gpu': int(cells), 'batches_in_flight': nfl,
                       'parallelism': 'pairs round-robin x%d' % world},
            'roofline': {'bound': 'hbm', 'achieved': round(achieved, 2), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': round(achieved / HBM_PEAK_GBS, 5), 'traffic': pmc['hbm_bytes_per_launch'] if pmc else None,
                         'counters_source': pmc_src,
                         # VALU wave-instructions per launch x 4 issue cycles over the SIMD-cycles of the live launch
                         # duration at the peak clock: how busy the vector ALUs are (the kernel's real bound)
                         'valu_busy_frac': round(valu * 4 / (N_SIMD * fill * 1e-3 * CLOCK_GHZ * 1e9), 4) if valu else None,
                         # lane-level VALU instructions per DP cell (wave-instructions x 64 lanes / cells; a packed
                         # instruction updates two cells, idle lanes included)
                         'valu_insts_per_cell': round(valu * 64 / cells, 3) if valu else None,
                         'valu_wave_insts_per_launch': valu,
                         'kernel': batch.kernel_name,
                         'kernel_ms': round(fill, 4),
                         'kernel_ms_note': 'HIP events on the launch stream, mean of %d launches of one batch running alone '
                                           '(serial leg); in the timed region %d fills co-run, each lasting %s ms'
                                           % (nx, nfl, '/'.join('%.2f' % v for v in overlapped_ms)),
                         'algorithmic_bytes_per_launch': int(alg_bytes),
                         'kernel_gcups': round(cells / (fill * 1e-3) / 1e9, 2),
                         'frac_timed_region': round(alg_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                         'traceback_kernel_ms': round(float(np.mean(trace_ms)), 4)},
            'results_ok': bool(ok), 'check': check,
        }
        line.update(extras)
        if variants:
            line['variants'] = variants
        line['cpu_baseline'] = cpu          # timed at N = 1 only (null otherwise, or with --no-cpu-baseline)
        if cpu_o2 is not None:
            line['cpu_baseline_O2'] = cpu_o2
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(line) + '\n').encode())
    for b in batches:
        b.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and not ok:
        sys.exit('bench.py: result check FAILED: %s' % json.dumps(check))


if __name__ == '__main__':
    main()
