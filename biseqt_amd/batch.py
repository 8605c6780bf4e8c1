"""Batches of independent alignment problems on one GPU (host side of include/pw_batch.h).

No reference counterpart: the reference solves one pair per ``Aligner`` (``biseqt/pw.py:119-319``).
Per pair a batch computes exactly what ``Aligner.solve()`` + ``Aligner.traceback()`` compute, for
all pairs in a handful of kernel launches.  Keyword arguments keep the reference's names
(``alnmode``, ``alntype``, ``subst_scores``, ``match_score``, ``mismatch_score``, ``go_score``,
``ge_score``, ``diag_range``).
"""
import ctypes as C

import numpy as np

from . import _pwlib as W
from .sequence import Sequence

RESULT_DTYPE = np.dtype([('score', '<f8'), ('opt_i', '<i4'), ('opt_j', '<i4'),
                         ('origin_idx', '<i4'), ('mutant_idx', '<i4'),
                         ('tx_len', '<i4'), ('status', '<i4')])
assert RESULT_DTYPE.itemsize == 32


def _as_u8(seq):
    if isinstance(seq, Sequence):
        return seq.as_array(np.uint8)
    a = np.asarray(seq)
    assert a.ndim == 1
    if a.size:
        assert a.min() >= 0 and a.max() <= 255, 'letters must be 0..255'
    return np.ascontiguousarray(a, dtype=np.uint8)


class DeviceBuffer(object):
    """A view of device memory owned by a batch, exportable to torch without a copy
    (``torch.as_tensor(buf, device='cuda')``) through ``__cuda_array_interface__``."""

    def __init__(self, ptr, nbytes, owner):
        self.ptr, self.nbytes, self.owner = ptr, nbytes, owner

    @property
    def __cuda_array_interface__(self):
        return dict(shape=(self.nbytes,), typestr='|u1', data=(self.ptr, False), version=2)


class PinnedArray(object):
    """Page-locked host memory (``pw_host_alloc``) viewed as a numpy array: the source / destination of the
    asynchronous transfers (``upload_async``, ``results_async``, ``transcripts_async``)."""

    def __init__(self, nbytes, dtype=np.uint8):
        self.lib = W.load()
        self.nbytes = int(nbytes)
        self.ptr = self.lib.pw_host_alloc(max(self.nbytes, 1))
        if not self.ptr:
            raise MemoryError('pw_host_alloc(%d) failed: %s' % (nbytes, W.last_error()))
        buf = (C.c_uint8 * max(self.nbytes, 1)).from_address(self.ptr)
        self.array = np.frombuffer(buf, dtype=np.uint8, count=self.nbytes).view(dtype)

    def close(self):
        if getattr(self, 'ptr', None):
            self.array = None
            self.lib.pw_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


PAIR_DTYPE = np.dtype([('origin_off', '<u8'), ('mutant_off', '<u8'), ('origin_len', '<i4'), ('mutant_len', '<i4'),
                       ('dmin', '<i4'), ('dmax', '<i4')])
assert PAIR_DTYPE.itemsize == 32


def pack_reads(reads):
    """All reads once in one letter arena, every read on a 16-byte boundary (frames must be 4-byte aligned, with
    slack behind): returns ``(arena uint8, offsets int64, lengths int32)`` for :meth:`BatchAligner.from_arena`."""
    arrs = [_as_u8(r) for r in reads]
    lens = np.array([len(a) for a in arrs], np.int64)
    sizes = (lens + 15) // 16 * 16 + 16
    offs = np.zeros(len(arrs), np.int64)
    if len(arrs):
        offs[1:] = np.cumsum(sizes)[:-1]
    arena = np.zeros(int(sizes.sum()) + 16, np.uint8)
    for a, o in zip(arrs, offs):
        arena[o:o + len(a)] = a
    return arena, offs, lens.astype(np.int32)


class DeviceArena(object):
    """One device copy of a read arena (:func:`pack_reads`) for all the batches of a job (``pw_arena_upload``)."""

    def __init__(self, arena, device=0):
        self.lib = W.load()
        arena = np.ascontiguousarray(arena, np.uint8)
        self.device, self.nbytes = device, arena.nbytes
        self.ptr = self.lib.pw_arena_upload(device, arena.ctypes.data, arena.nbytes)
        if not self.ptr:
            raise RuntimeError('pw_arena_upload failed: ' + W.last_error())

    def close(self):
        if self.ptr:
            self.lib.pw_arena_free(self.device, self.ptr)
            self.ptr = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BatchAligner(object):
    """Plan, upload, solve and trace back a batch of pairs.

    Args:
        pairs: iterable of ``(origin, mutant)``; each a :class:`Sequence` or an array of letter indices.
    Keyword Args:
        alphabet_len (int): number of letters (default: from the first Sequence, else max letter + 1).
        alnmode, alntype, subst_scores, match_score, mismatch_score, go_score, ge_score: as ``Aligner``.
        diag_range: one ``(dmin, dmax)`` for all pairs or a list with one per pair (banded mode).
        device (int): HIP device ordinal.  flags (int): ``PW_FLAG_*`` of include/pw_batch.h.
        check_band (bool): apply the reference's Python-side ``diag_range`` assertion (default True).
    """

    def __init__(self, pairs, **kw):
        self.lib = W.load()
        self.alnmode = kw.get('alnmode', W.STD_MODE)
        self.alntype = kw.get('alntype', W.GLOBAL)
        pairs = list(pairs)
        self.n = len(pairs)
        L = kw.get('alphabet_len')
        seqs = []
        for (o, m) in pairs:
            if L is None and isinstance(o, Sequence):
                L = len(o.alphabet)
            seqs.append((_as_u8(o), _as_u8(m)))
        if L is None:
            L = 1 + max([0] + [int(s.max()) for p in seqs for s in p if s.size])
        self.L = L
        subst = kw.get('subst_scores')
        if subst is None:
            match, mismatch = kw.get('match_score', 1), kw.get('mismatch_score', 0)
            subst = [[match if i == j else mismatch for i in range(L)] for j in range(L)]
        assert len(subst) == L
        self.subst_scores = subst
        self.go_score, self.ge_score = kw.get('go_score', 0), kw.get('ge_score', 0)
        dr = kw.get('diag_range')
        if self.alnmode == W.BANDED_MODE:
            assert dr is not None, 'banded mode needs diag_range'
            drs = [dr] * self.n if (len(dr) == 2 and not hasattr(dr[0], '__len__')) else list(dr)
            assert len(drs) == self.n
        else:
            drs = [(0, 0)] * self.n
        # arena: every sequence on a 16-byte boundary
        offs, total = [], 0
        for (o, m) in seqs:
            oo = total
            total += (len(o) + 15) // 16 * 16 + 16
            mo = total
            total += (len(m) + 15) // 16 * 16 + 16
            offs.append((oo, mo))
        self.arena = np.zeros(max(total, 16), np.uint8)
        self._pairs = (W.pw_pair * max(self.n, 1))()
        for k, ((o, m), (oo, mo), (dmin, dmax)) in enumerate(zip(seqs, offs, drs)):
            self.arena[oo:oo + len(o)] = o
            self.arena[mo:mo + len(m)] = m
            if self.alnmode == W.BANDED_MODE and kw.get('check_band', True):
                # the reference's Python-side check (pw.py:224-226); check_band=False hands the band to
                # the C side unchecked, which clamps / rejects it like dptable_init
                assert -len(m) <= dmin <= dmax <= len(o), 'diag_range outside the table'
            self._pairs[k] = W.pw_pair(oo, mo, len(o), len(m), int(dmin), int(dmax))
        self.lens = [(len(o), len(m)) for (o, m) in seqs]
        S = np.ascontiguousarray(np.asarray(subst, dtype=np.float64).reshape(L, L))
        self._S = S
        sc = W.pw_scoring(self.alnmode, self.alntype, L, S.ctypes.data_as(C.POINTER(C.c_double)),
                          float(self.go_score), float(self.ge_score))
        self.device = kw.get('device', 0)
        self.flags = kw.get('flags', 0)
        self.handle = self.lib.pw_batch_create(self.device, C.byref(sc), self.n, self._pairs,
                                               self.arena.nbytes, self.flags)
        if not self.handle:
            raise RuntimeError('pw_batch_create failed: ' + W.last_error())
        if kw.get('upload', True):
            self.upload()

    @classmethod
    def from_arena(cls, arena, offsets, lengths, pairs, diag_ranges=None, device_arena=None, **kw):
        """Pairs that REFER to reads of one shared arena (:func:`pack_reads`) instead of carrying copies: ``pairs`` is
        an (n, 2) array of read indices (origin, mutant), ``diag_ranges`` an (n, 2) array in banded mode.  This is the
        shape of overlap pipelines, where every read takes part in dozens of pairs.  With ``device_arena`` (a
        :class:`DeviceArena` of the same arena) the batch reads the reads from that device copy -- uploaded once for
        all the batches of a job -- instead of allocating and uploading its own."""
        self = cls.__new__(cls)
        self.lib = W.load()
        self.alnmode = kw.get('alnmode', W.STD_MODE)
        self.alntype = kw.get('alntype', W.GLOBAL)
        pairs = np.ascontiguousarray(pairs, np.int64).reshape(-1, 2)
        self.n = len(pairs)
        L = kw['alphabet_len']
        self.L = L
        subst = kw.get('subst_scores')
        if subst is None:
            match, mismatch = kw.get('match_score', 1), kw.get('mismatch_score', 0)
            subst = [[match if i == j else mismatch for i in range(L)] for j in range(L)]
        self.subst_scores = subst
        self.go_score, self.ge_score = kw.get('go_score', 0), kw.get('ge_score', 0)
        offsets, lengths = np.asarray(offsets, np.int64), np.asarray(lengths, np.int64)
        assert (offsets % 4 == 0).all(), 'reads must start on 4-byte boundaries'
        rec = np.zeros(max(self.n, 1), PAIR_DTYPE)
        rec['origin_off'][:self.n] = offsets[pairs[:, 0]]; rec['mutant_off'][:self.n] = offsets[pairs[:, 1]]
        rec['origin_len'][:self.n] = lengths[pairs[:, 0]]; rec['mutant_len'][:self.n] = lengths[pairs[:, 1]]
        if self.alnmode == W.BANDED_MODE:
            dr = np.asarray(diag_ranges, np.int64).reshape(-1, 2)
            assert len(dr) == self.n
            if kw.get('check_band', True):
                assert ((-rec['mutant_len'][:self.n] <= dr[:, 0]) & (dr[:, 0] <= dr[:, 1]) &
                        (dr[:, 1] <= rec['origin_len'][:self.n])).all(), 'diag_range outside the table'
            rec['dmin'][:self.n] = dr[:, 0]; rec['dmax'][:self.n] = dr[:, 1]
        self._pairs = rec
        self.arena = np.ascontiguousarray(arena, np.uint8)
        self.lens = None
        S = np.ascontiguousarray(np.asarray(subst, dtype=np.float64).reshape(L, L))
        self._S = S
        sc = W.pw_scoring(self.alnmode, self.alntype, L, S.ctypes.data_as(C.POINTER(C.c_double)),
                          float(self.go_score), float(self.ge_score))
        self.device = kw.get('device', 0)
        self.flags = kw.get('flags', 0) | (W.PW_FLAG_SHARED_ARENA if device_arena is not None else 0)
        self.handle = self.lib.pw_batch_create(self.device, C.byref(sc), self.n, rec.ctypes.data_as(C.POINTER(W.pw_pair)),
                                               self.arena.nbytes, self.flags)
        if not self.handle:
            raise RuntimeError('pw_batch_create failed: ' + W.last_error())
        if device_arena is not None:
            assert device_arena.device == self.device and device_arena.nbytes >= self.arena.nbytes
            self._device_arena = device_arena          # keeps it alive as long as the batch
            self._ck(self.lib.pw_batch_share_arena(self.handle, device_arena.ptr), 'pw_batch_share_arena')
        else:
            self.upload()
        return self

    # ---- lifecycle ----
    def close(self):
        if getattr(self, 'handle', None):
            self.lib.pw_batch_destroy(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc, what):
        if rc != 0:
            raise RuntimeError('%s failed: %s' % (what, W.last_error()))

    # ---- planning info ----
    @property
    def cells(self):
        """Cells the reference would allocate for the batch: the GCUPS denominator."""
        return self.lib.pw_batch_cells(self.handle)

    @property
    def algorithmic_bytes(self):
        return self.lib.pw_batch_algorithmic_bytes(self.handle)

    @property
    def score_dtype(self):
        return 'f64' if self.lib.pw_batch_score_type(self.handle) else 'i32'

    @property
    def kernel_name(self):
        return self.lib.pw_batch_kernel_name(self.handle).decode()

    def init_rc(self, k):
        return self.lib.pw_batch_init_rc(self.handle, k)

    def band(self, k):
        a, b, r = C.c_int32(), C.c_int32(), C.c_int32()
        self.lib.pw_batch_band(self.handle, k, C.byref(a), C.byref(b), C.byref(r))
        return a.value, b.value, r.value

    # ---- data movement and kernels ----
    def upload(self):
        self._ck(self.lib.pw_batch_upload_arena(self.handle, self.arena.ctypes.data, self.arena.nbytes),
                 'pw_batch_upload_arena')

    def upload_async(self, pinned, stream=None):
        """H2D of the arena from a :class:`PinnedArray` holding it, asynchronous on ``stream``."""
        self._ck(self.lib.pw_batch_upload_arena_async(self.handle, pinned.ptr, self.arena.nbytes, stream),
                 'pw_batch_upload_arena_async')

    def results_async(self, pinned, stream=None):
        """D2H of the 32-byte records into a :class:`PinnedArray` (``32 * n`` bytes), asynchronous on ``stream``."""
        self._ck(self.lib.pw_batch_results_async(self.handle, pinned.ptr, stream), 'pw_batch_results_async')

    def transcripts_async(self, pinned, stream=None):
        """D2H of all transcript slots into a :class:`PinnedArray` (``transcripts_bytes`` bytes), asynchronous."""
        self._ck(self.lib.pw_batch_transcripts_async(self.handle, pinned.ptr, stream), 'pw_batch_transcripts_async')

    @property
    def transcripts_bytes(self):
        return int(self.lib.pw_batch_transcripts_bytes(self.handle))

    def solve(self, stream=None):
        self._ck(self.lib.pw_batch_solve(self.handle, stream), 'pw_batch_solve')

    def traceback(self, stream=None):
        self._ck(self.lib.pw_batch_traceback(self.handle, stream), 'pw_batch_traceback')

    def traceback_from(self, ends, stream=None):
        e = np.ascontiguousarray(np.asarray(ends, dtype=np.int32).reshape(self.n, 2))
        self._ck(self.lib.pw_batch_traceback_from(self.handle, e.ctypes.data_as(C.POINTER(C.c_int32)), stream),
                 'pw_batch_traceback_from')

    def sync(self, stream=None):
        self._ck(self.lib.pw_batch_sync(self.handle, stream), 'pw_batch_sync')

    def run(self, stream=None):
        """solve + traceback + sync; returns ``results()``."""
        self.solve(stream)
        self.traceback(stream)
        self.sync(stream)
        return self.results()

    # ---- results ----
    def results(self):
        """Structured array (RESULT_DTYPE), one record per pair (synchronous D2H)."""
        out = np.zeros(max(self.n, 1), RESULT_DTYPE)
        self._ck(self.lib.pw_batch_results(self.handle, out.ctypes.data), 'pw_batch_results')
        return out[:self.n]

    def transcripts(self, results=None, slots=None):
        """List with one transcript string (or None) per pair (synchronous D2H, unless ``slots`` already holds the
        slot buffer, e.g. from :meth:`transcripts_async`)."""
        res = self.results() if results is None else results
        nb = self.lib.pw_batch_transcripts_bytes(self.handle)
        if slots is not None:
            buf = slots
        else:
            buf = np.zeros(max(nb, 1), np.uint8)
            self._ck(self.lib.pw_batch_transcripts(self.handle, buf.ctypes.data), 'pw_batch_transcripts')
        out = []
        off, cap = C.c_uint64(), C.c_int32()
        for k in range(self.n):
            n = int(res['tx_len'][k])
            if n <= 0:
                out.append(None)
                continue
            self.lib.pw_batch_tx_slot(self.handle, k, C.byref(off), C.byref(cap))
            end = off.value + cap.value
            out.append(buf[end - n:end].tobytes().decode('ascii'))
        return out

    def transcript_slots(self):
        """(offsets, capacities) of all transcript slots as arrays (ops of pair k END at offset + capacity)."""
        off, cap = np.zeros(self.n, np.uint64), np.zeros(self.n, np.int32)
        o, c = C.c_uint64(), C.c_int32()
        for k in range(self.n):
            self.lib.pw_batch_tx_slot(self.handle, k, C.byref(o), C.byref(c))
            off[k], cap[k] = o.value, c.value
        return off, cap

    # ---- compacted transcripts (what a gather should move) ----
    def pack_transcripts(self, stream=None):
        """After a traceback on ``stream``: the ops of all pairs back to back in pair order and their exclusive offsets
        (uint64[n + 1]), device-resident (``pw_batch_pack_transcripts``).  Asynchronous on ``stream``."""
        self._ck(self.lib.pw_batch_pack_transcripts(self.handle, stream), 'pw_batch_pack_transcripts')

    def packed_device(self):
        """(packed bytes, offsets) as :class:`DeviceBuffer` views; the packed view spans the whole buffer (its first
        ``offsets[n]`` bytes are the ops)."""
        return (DeviceBuffer(self.lib.pw_batch_packed_device(self.handle), max(self.transcripts_bytes, 16), self),
                DeviceBuffer(self.lib.pw_batch_packed_offsets_device(self.handle), 8 * (self.n + 1), self))

    def packed_total_async(self, pinned, stream=None):
        """D2H of the total number of packed bytes into a :class:`PinnedArray` of 8 bytes, ordered on ``stream``."""
        self._ck(self.lib.pw_batch_packed_total_async(self.handle, pinned.ptr, stream), 'pw_batch_packed_total_async')

    def packed(self):
        """(bytes uint8[total], offsets uint64[n + 1]) on the host (synchronous)."""
        off = np.zeros(self.n + 1, np.uint64)
        self._ck(self.lib.pw_batch_packed(self.handle, None, 0, off.ctypes.data), 'pw_batch_packed')
        buf = np.zeros(max(int(off[-1]), 1), np.uint8)
        self._ck(self.lib.pw_batch_packed(self.handle, buf.ctypes.data, buf.nbytes, None), 'pw_batch_packed')
        return buf[:int(off[-1])], off

    @staticmethod
    def transcripts_from_packed(buf, offsets):
        """List of transcript strings (None for pairs without one) from a packed buffer and its offsets."""
        offsets = np.asarray(offsets, np.int64)
        raw = np.asarray(buf, np.uint8).tobytes()
        return [raw[offsets[k]:offsets[k + 1]].decode('ascii') if offsets[k + 1] > offsets[k] else None
                for k in range(len(offsets) - 1)]

    def scores_plane(self, k):
        """Score of every cell of pair k as ``plane[d - dmin, a]`` (needs PW_FLAG_DUMP_SCORES)."""
        X, Y = self.lens[k]
        dmin, dmax, _ = self.band(k)
        nd, pitch = dmax - dmin + 1, min(X, Y) + 1
        out = np.zeros(nd * pitch, np.float64)
        self._ck(self.lib.pw_batch_scores(self.handle, k, out.ctypes.data_as(C.POINTER(C.c_double)), out.size),
                 'pw_batch_scores')
        return out.reshape(nd, pitch)

    def results_device(self):
        return DeviceBuffer(self.lib.pw_batch_results_device(self.handle), 32 * self.n, self)

    def transcripts_device(self):
        return DeviceBuffer(self.lib.pw_batch_transcripts_device(self.handle),
                            self.lib.pw_batch_transcripts_bytes(self.handle), self)

    def fill_ms(self):
        return float(self.lib.pw_batch_fill_ms(self.handle))

    def trace_ms(self):
        return float(self.lib.pw_batch_trace_ms(self.handle))


def align_batch(pairs, **kw):
    """One-call convenience: returns ``(results, transcripts)`` for the pairs."""
    with BatchAligner(pairs, **kw) as b:
        res = b.run()
        return res, b.transcripts(res)


def plan_only(shapes, alnmode=W.STD_MODE, alntype=W.GLOBAL, alphabet_len=4, subst_scores=None, match_score=1, mismatch_score=0,
              go_score=0, ge_score=0, flags=0):
    """What :class:`BatchAligner` would choose for problems of these shapes, without a GPU and without allocating anything
    (``pw_plan_only``): ``shapes`` is a list of ``(origin_len, mutant_len)`` or ``(origin_len, mutant_len, dmin, dmax)``.
    Returns a dict: ``kernel`` (as :attr:`BatchAligner.kernel_name`), ``score_dtype``, ``scale_shift`` (dyadic scaling:
    scores held times ``2**shift``), the number of pairs on one-wavefront kernels / workgroup kernels / the tiled kernel /
    the strip pipeline, ``packed_rule`` (-1 unless a packed 16-bit kernel) and ``matrix`` (packed kernel in matrix form)."""
    lib = W.load()
    L = alphabet_len
    if subst_scores is None:
        subst_scores = [[match_score if i == j else mismatch_score for i in range(L)] for j in range(L)]
    S = np.ascontiguousarray(np.asarray(subst_scores, dtype=np.float64).reshape(L, L))
    sc = W.pw_scoring(alnmode, alntype, L, S.ctypes.data_as(C.POINTER(C.c_double)), float(go_score), float(ge_score))
    pairs = (W.pw_pair * max(len(shapes), 1))()
    off = 0
    for k, sh in enumerate(shapes):
        X, Y = int(sh[0]), int(sh[1])
        dmin, dmax = (int(sh[2]), int(sh[3])) if len(sh) > 2 else (0, 0)
        oo = off
        off += (X + 15) // 16 * 16 + 16
        mo = off
        off += (Y + 15) // 16 * 16 + 16
        pairs[k] = W.pw_pair(oo, mo, X, Y, dmin, dmax)
    name = C.create_string_buffer(128)
    info = (C.c_int32 * 8)()
    if lib.pw_plan_only(C.byref(sc), len(shapes), pairs, max(off, 16), flags, name, 128, info) != 0:
        raise RuntimeError('pw_plan_only failed: ' + W.last_error())
    return dict(kernel=name.value.decode(), score_dtype='f64' if info[0] else 'i32', scale_shift=info[1], one_wavefront=info[2],
                workgroup=info[3], tiled=info[4], strips=info[5], packed_rule=info[6], matrix=bool(info[7]))
