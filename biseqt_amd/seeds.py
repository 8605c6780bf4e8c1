"""Exact-match k-mer seeds in diagonal coordinates (host side of include/pw_seeds.h).

Mirrors ``biseqt/seeds.py:SeedIndex`` (:21-237): same constructor keywords (``wordlen``, ``alphabet``, ``mask``;
``path`` / ``kmer_cache`` / ``log_level`` are accepted and ignored -- there is no SQLite file, the table lives in
HBM), same classmethods for the coordinate maps, same ``seeds()`` / ``seed_count()`` results in the same order.

    >>> from biseqt_amd.seeds import SeedIndex
    >>> from biseqt_amd.sequence import Alphabet
    >>> A = Alphabet('ACGT')
    >>> S, T = A.parse('TAAGCGT'), A.parse('GGCGTAA')
    >>> list(SeedIndex(S, T, wordlen=3, alphabet=A).seeds())
    [(4, 2), (3, 1), (0, 4)]
"""
import ctypes as C
from itertools import product

import numpy as np

from . import _pwlib as W
from .batch import DeviceBuffer
from .kmers import check_limits, mask_bits
from .sequence import Sequence


class _Index(object):
    """Thin owner of a ``pw_seed_index`` handle."""

    def __init__(self, S, T, wordlen, alphabet, mask=(), self_comp=-1, device=0):
        self.lib = W.load()
        s = S.as_array(np.uint8) if isinstance(S, Sequence) else np.ascontiguousarray(S, np.uint8)
        t = T.as_array(np.uint8) if isinstance(T, Sequence) else np.ascontiguousarray(T, np.uint8)
        bits = mask_bits(list(mask))
        marr = (C.c_uint64 * max(len(bits), 1))(*bits)
        self.nS, self.nT, self.wordlen = len(s), len(t), wordlen
        self.handle = self.lib.pw_seeds_create(device, s.ctypes.data, len(s), t.ctypes.data, len(t), len(alphabet),
                                               wordlen, marr, len(bits), self_comp)
        if not self.handle:
            raise RuntimeError('pw_seeds_create failed: ' + self.error())
        self.built = False

    def error(self):
        return (self.lib.pw_seeds_last_error() or b'').decode('utf-8', 'replace')

    def build(self, max_rows=0, stream=None):
        if self.lib.pw_seeds_build(self.handle, max_rows, stream) != 0:
            raise RuntimeError('pw_seeds_build failed: ' + self.error())
        self.built = True
        return self.lib.pw_seeds_num_rows(self.handle)

    def rows(self):
        """(n, 2) int32 array of (d, a) in table order."""
        n = self.lib.pw_seeds_num_rows(self.handle)
        out = np.zeros((max(n, 1), 2), np.int32)
        if self.lib.pw_seeds_rows(self.handle, out.ctypes.data, n) != 0:
            raise RuntimeError('pw_seeds_rows failed: ' + self.error())
        return out[:n]

    def rows_device(self):
        n = self.lib.pw_seeds_num_rows(self.handle)
        return DeviceBuffer(self.lib.pw_seeds_rows_device(self.handle), 8 * n, self)

    def count(self, d_band=None, a_band=None):
        d = d_band if d_band is not None else (0, 0)
        a = a_band if a_band is not None else (0, 0)
        n = self.lib.pw_seeds_count(self.handle, d_band is not None, int(d[0]), int(d[1]),
                                    a_band is not None, int(a[0]), int(a[1]))
        if n < 0:
            raise RuntimeError('pw_seeds_count failed: ' + self.error())
        return n

    def band_neighbours(self, radius):
        """Per row: how many other rows lie within 1 on the axis d / radius(d) (radius: table over d = -nT .. nS)."""
        n = self.lib.pw_seeds_num_rows(self.handle)
        rad = np.ascontiguousarray(radius, np.float64)
        out = np.zeros(max(n, 1), np.int32)
        if self.lib.pw_seeds_band_neighbours(self.handle, rad.ctypes.data, rad.size, out.ctypes.data, n) != 0:
            raise RuntimeError('pw_seeds_band_neighbours failed: ' + self.error())
        return out[:n]

    def graph_build(self, d_coeff, radius):
        """Neighbourhood graph of the rows: max(|d - d'| * d_coeff, |a - a'|) <= radius.  Returns the edge count."""
        e = self.lib.pw_seeds_graph_build(self.handle, float(d_coeff), float(radius))
        if e < 0:
            raise RuntimeError('pw_seeds_graph_build failed: ' + self.error())
        self._edges = e
        return e

    def graph_points(self):
        """(n, 2) int32 array of the graph's points (d, a): the rows, or for a self comparison the non-trivial rows
        each followed by its mirror image."""
        n = self.lib.pw_seeds_graph_num_points(self.handle)
        out = np.zeros((max(n, 1), 2), np.int32)
        if self.lib.pw_seeds_graph_points(self.handle, out.ctypes.data, n) != 0:
            raise RuntimeError('pw_seeds_graph_points failed: ' + self.error())
        return out[:n]

    def graph_counts(self):
        n = self.lib.pw_seeds_graph_num_points(self.handle)
        out = np.zeros(max(n, 1), np.int32)
        if self.lib.pw_seeds_graph_counts(self.handle, out.ctypes.data, n) != 0:
            raise RuntimeError('pw_seeds_graph_counts failed: ' + self.error())
        return out[:n]

    def graph_fetch(self):
        """CSR adjacency: (offsets[n + 1], neighbours[edges])."""
        n = self.lib.pw_seeds_graph_num_points(self.handle)
        off = np.zeros(n + 1, np.int64)
        adj = np.zeros(max(self._edges, 1), np.int32)
        if self.lib.pw_seeds_graph_fetch(self.handle, off.ctypes.data, adj.ctypes.data) != 0:
            raise RuntimeError('pw_seeds_graph_fetch failed: ' + self.error())
        return off, adj[:self._edges]

    def graph_components(self, avail):
        n = self.lib.pw_seeds_graph_num_points(self.handle)
        av = np.ascontiguousarray(avail, np.uint8)
        assert av.size == n
        out = np.full(max(n, 1), -1, np.int32)
        if self.lib.pw_seeds_graph_components(self.handle, av.ctypes.data, out.ctypes.data) != 0:
            raise RuntimeError('pw_seeds_graph_components failed: ' + self.error())
        return out[:n]

    def kmers(self, which):
        n = (self.nT if which else self.nS) - self.wordlen + 1
        out = np.zeros(max(n, 1), np.int64)
        got = self.lib.pw_seeds_kmers(self.handle, which, out.ctypes.data, max(n, 0))
        if got < 0:
            raise RuntimeError('pw_seeds_kmers failed: ' + self.error())
        return out[:got]

    @property
    def is_self(self):
        return bool(self.lib.pw_seeds_is_self(self.handle))

    def build_ms(self):
        return self.lib.pw_seeds_build_ms(self.handle)

    def algorithmic_bytes(self):
        return self.lib.pw_seeds_algorithmic_bytes(self.handle)

    def close(self):
        if getattr(self, 'handle', None):
            self.lib.pw_seeds_destroy(self.handle)
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


class SeedIndex(object):
    """An index for seeds in diagonal coordinates (``seeds.py:21-47``).

    Args:
        S, T (Sequence): the 1st and 2nd sequence; equal contents make it a self comparison.
    Keyword Args:
        wordlen (int), alphabet (Alphabet), mask (list of sets): as in the reference.
        device (int): HIP device ordinal.  max_rows (int): refuse tables larger than this (default 2^31 - 1).
    """

    def __init__(self, S, T, kmer_cache=None, **kw):
        alphabet, wordlen = kw['alphabet'], kw['wordlen']
        check_limits(alphabet, wordlen)
        assert isinstance(S, Sequence) and isinstance(T, Sequence)
        assert S.alphabet == alphabet and T.alphabet == alphabet
        self.alphabet, self.wordlen = alphabet, wordlen
        self.mask = kw.get('mask', [])
        self.S, self.T = S, T
        self.self_comp = S == T                               # seeds.py:33
        self._idx = _Index(S, T, wordlen, alphabet, self.mask, self_comp=int(self.self_comp),
                           device=kw.get('device', 0))
        self._idx.build(kw.get('max_rows', 0))
        self._rows = None

    # ---- coordinate maps (seeds.py:55-106) ----
    @classmethod
    def to_diagonal_coordinates(cls, i, j):
        return i - j, i + j

    @classmethod
    def to_ij_coordinates(cls, d, a):
        # the reference divides with python 2's integer `/`; a + d and a - d are even for every seed
        return (a + d) // 2, (a - d) // 2

    @classmethod
    def to_ij_coordinates_seg(cls, seg):
        corners = [cls.to_ij_coordinates(d, a) for d, a in product(*seg)]
        i_start = max(min(i for i, _ in corners), 0)
        j_start = max(min(j for _, j in corners), 0)
        i_end = max(i for i, j in corners)
        j_end = max(j for _, j in corners)
        return (i_start, i_end), (j_start, j_end)

    # ---- the table ----
    def rows(self):
        """(n, 2) int32 array of (d, a) in the reference's rowid order (cached)."""
        if self._rows is None:
            self._rows = self._idx.rows()
        return self._rows

    def rows_device(self):
        return self._idx.rows_device()

    def seeds(self, d_band=None, exclude_trivial=False):
        """Yields all seeds ``(i, j)``, optionally those in a diagonal band (``seeds.py:164-197``); a self
        comparison also yields the mirror image of every non-trivial seed."""
        rows = self.rows()
        if d_band is not None:
            assert len(d_band) == 2, 'need a 2-tuple for diagonal band'
            rows = rows[(rows[:, 0] >= d_band[0]) & (rows[:, 0] <= d_band[1])]
        d, a = rows[:, 0].astype(np.int64), rows[:, 1].astype(np.int64)
        ii, jj = ((a + d) // 2).tolist(), ((a - d) // 2).tolist()
        for i, j in zip(ii, jj):
            if self.self_comp and exclude_trivial and i == j:
                continue
            yield (i, j)
            if self.self_comp and i != j:
                yield (j, i)

    def seed_count(self, d_band=None, a_band=None):
        """Number of rows of the table, optionally inside a diagonal and / or an antidiagonal band
        (``seeds.py:199-237``); counted on the device."""
        if d_band is not None:
            assert len(d_band) == 2, 'need a 2-tuple for diagonal band'
        if a_band is not None:
            assert len(a_band) == 2, 'need a 2-tuple for antidiagonal band'
        return self._idx.count(d_band, a_band)

    def build_ms(self):
        return self._idx.build_ms()

    def close(self):
        self._idx.close()
