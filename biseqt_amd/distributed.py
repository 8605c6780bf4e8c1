"""Multi-GPU batches: one process per GPU, pairs dealt round-robin, one gather of the results.

Independent sequence pairs are the natural shard: pair ``p`` goes to rank
``p mod world``; for batches of uneven pairs the list is first sorted by table size so the deal is
balanced.  There is no collective on the data path; the only exchange is the final gather of the
fixed 32-byte result records (and, optionally, the transcripts) to rank 0 through
``torch.distributed`` -- backend ``nccl`` (= RCCL over xGMI) on GPUs, ``gloo`` in the CPU tests.
The traffic is KBs-MBs and latency-bound: every rank talks to the root over its own direct xGMI link,
so a plain gather (not a ring) is the right shape.
"""
import numpy as np

from .batch import RESULT_DTYPE


def shard_indices(n_pairs, rank, world, weights=None):
    """Indices of the pairs rank ``rank`` owns.  Round-robin; with ``weights`` (e.g. cells per pair)
    the pairs are dealt heaviest first so every rank gets a similar load."""
    if weights is None:
        return np.arange(rank, n_pairs, world)
    order = np.argsort(-np.asarray(weights), kind='stable')
    return np.sort(order[rank::world])


def gather_records(local_records, n_total, rank, world, owner_of=None, device=None, group=None):
    """Gather per-rank result records to rank 0.

    ``local_records``: RESULT_DTYPE array (host) or a uint8 torch tensor of 32*n_local bytes that may
    live on the GPU.  Returns, on rank 0, a RESULT_DTYPE array of ``n_total`` records in global pair
    order (``owner_of(rank)`` gives the global indices a rank owns; default round-robin); None elsewhere.
    """
    import torch
    import torch.distributed as dist
    if owner_of is None:
        def owner_of(r):
            return np.arange(r, n_total, world)
    counts = [len(owner_of(r)) for r in range(world)]
    cap = max(counts + [1])
    if isinstance(local_records, np.ndarray):
        t = torch.from_numpy(local_records.view(np.uint8).reshape(-1).copy())
        if device is not None:
            t = t.to(device)
    else:
        t = local_records.reshape(-1)
    pad = torch.zeros(cap * 32, dtype=torch.uint8, device=t.device)
    pad[:t.numel()] = t
    if world == 1:
        chunks = [pad]
    else:
        # a true gather: rank r -> rank 0 only (RCCL implements it as grouped send/recv over each rank's
        # direct xGMI link to the root; gloo has it natively)
        chunks = [torch.empty_like(pad) for _ in range(world)] if rank == 0 else None
        dist.gather(pad, chunks, dst=0, group=group)
    if rank != 0:
        return None
    full = np.zeros(n_total, RESULT_DTYPE)
    for r in range(world):
        rec = chunks[r].cpu().numpy()[:counts[r] * 32].view(RESULT_DTYPE)
        full[owner_of(r)] = rec
    return full


def gather_struct(local_records, n_total, rank, world, owner_of=None, device=None, group=None):
    """:func:`gather_records` for any fixed-size record type (e.g. the 64-byte overlap-band records of
    ``include/pw_overlap.h``): ``local_records`` is a numpy structured array; returns, on rank 0, the ``n_total``
    records in global order, None elsewhere."""
    import torch
    import torch.distributed as dist
    dt = local_records.dtype
    size = dt.itemsize
    if owner_of is None:
        def owner_of(r):
            return np.arange(r, n_total, world)
    counts = [len(owner_of(r)) for r in range(world)]
    cap = max(counts + [1])
    t = torch.from_numpy(np.ascontiguousarray(local_records).view(np.uint8).reshape(-1).copy())
    if device is not None:
        t = t.to(device)
    pad = torch.zeros(cap * size, dtype=torch.uint8, device=t.device)
    pad[:t.numel()] = t
    if world == 1:
        chunks = [pad]
    else:
        chunks = [torch.empty_like(pad) for _ in range(world)] if rank == 0 else None
        dist.gather(pad, chunks, dst=0, group=group)
    if rank != 0:
        return None
    full = np.zeros(n_total, dt)
    for r in range(world):
        full[owner_of(r)] = chunks[r].cpu().numpy()[:counts[r] * size].view(dt)
    return full


def gather_bytes(local_bytes, rank, world, group=None):
    """Gather one ragged uint8 tensor per rank to rank 0 (transcripts): sizes first, then padded
    payloads.  Returns the list of per-rank tensors on rank 0, None elsewhere."""
    import torch
    import torch.distributed as dist
    n = torch.tensor([local_bytes.numel()], dtype=torch.int64, device=local_bytes.device)
    if world == 1:
        return [local_bytes]
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    cap = max(int(s.item()) for s in sizes)
    pad = torch.zeros(max(cap, 1), dtype=torch.uint8, device=local_bytes.device)
    pad[:local_bytes.numel()] = local_bytes
    chunks = [torch.empty_like(pad) for _ in range(world)] if rank == 0 else None
    dist.gather(pad, chunks, dst=0, group=group)
    if rank != 0:
        return None
    return [chunks[r][:int(sizes[r].item())] for r in range(world)]


def gather_ragged_start(local, recv, rank, world, group=None):
    """The transcript gather of a step, with sizes agreed beforehand: rank r > 0 sends its uint8 tensor ``local``,
    rank 0 receives it into the pre-allocated ``recv[r]`` (``recv[0]`` is not used: rank 0 keeps its own).  One grouped
    send / receive -- the way RCCL itself builds a gather, each rank over its direct xGMI link to the root -- with no
    size exchange, no padding copy and no host synchronisation.  Returns the requests; :func:`gather_ragged_wait`
    orders the current stream behind them."""
    import torch.distributed as dist
    if world == 1:
        return []
    if rank == 0:
        ops = [dist.P2POp(dist.irecv, recv[r], r, group) for r in range(1, world)]
    else:
        ops = [dist.P2POp(dist.isend, local, 0, group)]
    return dist.batch_isend_irecv(ops)


def gather_ragged_wait(reqs):
    for r in reqs:
        r.wait()


class DelayedRaggedGather(object):
    """Gather of one ragged uint8 payload per rank and step to rank 0 when the payload's SIZE is only known on the device
    at the time the step is issued -- the packed transcripts (``pw_batch_pack_transcripts``): their byte count is the last
    entry of an exclusive scan the GPU has not run yet.  Waiting for it would put a host synchronisation into every step,
    so the payload travels one use of its slot later, when the size has long arrived on the host:

      step i, slot j (``post``):     the fixed-size traffic of the step goes out as usual -- result records, and the 8-byte
                                     byte count of every rank (a plain ``gather`` the caller issues: ``totals_gathered``);
                                     the payload and a way to read its size on the host are remembered
      next use of slot j, or drain   (``collect``): the sender reads its own count (an 8-byte D2H the stream finished long
                                     ago), the root reads the gathered counts (one small D2H), and ONE grouped send / receive
                                     moves exactly the bytes that exist -- no padding, each rank over its direct link to the root

    ``caps[r]`` is the largest payload rank r can ever send (the slot bytes); rank 0 allocates its receive buffers once."""

    def __init__(self, rank, world, caps, device=None, group=None):
        import torch
        self.rank, self.world, self.group = rank, world, group
        self.recv = None
        if rank == 0 and world > 1:
            self.recv = [None] + [torch.empty(max(int(caps[r]), 1), dtype=torch.uint8, device=device) for r in range(1, world)]
        self.pending = None
        self.last = None

    def post(self, payload, own_total, totals_gathered=None):
        """``payload``: uint8 tensor holding this rank's bytes in its first ``own_total()`` bytes (callable, evaluated at
        collect time); ``totals_gathered``: on rank 0 the int64 tensor [world] of every rank's count, filled by the
        fixed-size gather of the same step."""
        assert self.pending is None, 'collect() the previous step of this slot first'
        self.pending = (payload, own_total, totals_gathered)

    def collect(self):
        """Moves the pending payloads; on rank 0 returns the list of per-rank uint8 tensors (rank 0's own first), None on
        the other ranks and when nothing is pending."""
        import torch.distributed as dist
        if self.pending is None:
            return None
        payload, own_total, totals = self.pending
        self.pending = None
        n_own = int(own_total())
        if self.world == 1:
            self.last = [payload[:n_own]]
            return self.last
        if self.rank == 0:
            sizes = [int(v) for v in totals.cpu().tolist()]
            assert sizes[0] == n_own, (sizes[0], n_own)
            ops = [dist.P2POp(dist.irecv, self.recv[r][:sizes[r]], r, self.group) for r in range(1, self.world) if sizes[r] > 0]
        else:
            ops = [dist.P2POp(dist.isend, payload[:n_own], 0, self.group)] if n_own > 0 else []
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        if self.rank != 0:
            return None
        self.last = [payload[:n_own]] + [self.recv[r][:sizes[r]] for r in range(1, self.world)]
        return self.last


def exchange_sizes(n_local, rank, world, device=None, group=None):
    """Every rank's byte count, as a list of ints on every rank (once, at set-up time)."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return [int(n_local)]
    t = torch.tensor([int(n_local)], dtype=torch.int64, device=device)
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t, group=group)
    return [int(o.item()) for o in out]
