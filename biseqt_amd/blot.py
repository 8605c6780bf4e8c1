"""Band selection for overlap alignments (host side; mirrors ``biseqt/blot.py``).

What is here: the closed-form geometry and statistics of Word-Blot (``wall_to_wall_distance`` :78-89,
``expected_overlap_len`` :92-112, ``band_radius`` / ``band_radii`` :116-160, ``H0_moments`` / ``H1_moments``
:163-218, ``find_peaks`` :37-76) -- scalar arithmetic, evaluated on the host exactly as the reference writes it --
``WordBlot`` (:228-490: ``score_seeds``, ``similar_segments``) and ``WordBlotOverlap`` (:490-579), whose heavy
parts -- for every seed the seeds in its neighbourhood (KD-tree ball queries in the reference) and the growth of
neighbouring seeds into segments (a depth-first search there) -- run on the GPU (kernels K6, K7 of pw_seeds.hip).
``highest_scoring_overlap_band()`` returns the ``d_band`` that the banded overlap aligner
(``Aligner(..., alnmode=BANDED_MODE, alntype=B_OVERLAP, diag_range=d_band)``) is given.
"""
import numpy as np
from scipy.special import erfcinv

from .seeds import SeedIndex


def find_peaks(xs, rs, threshold):
    """Maximal disjoint bands ``(i - 1, i + 1)`` around positions with ``xs[i] >= threshold``; bands closer than
    the radius are merged (``blot.py:37-76``)."""
    peaks, cur_peak = [], None
    for idx, x in enumerate(xs):
        radius = rs[idx] if isinstance(rs, (list, tuple, np.ndarray)) else rs
        if x < threshold:
            continue
        peak_l, peak_r = max(0, idx - 1), min(len(xs) - 1, idx + 1)
        if cur_peak is None:
            cur_peak = (peak_l, peak_r)
            continue
        if peak_l < cur_peak[1] + radius:
            assert peak_r >= cur_peak[1]
            cur_peak = (cur_peak[0], peak_r)
        else:
            peaks.append(cur_peak)
            cur_peak = (peak_l, peak_r)
    if cur_peak is not None:
        peaks.append(cur_peak)
    return [(int(l), int(r)) for (l, r) in peaks]


def wall_to_wall_distance(len0, len1, diag):
    return min(len0 - diag, len1) + min(diag, 0)


def expected_overlap_len(len0, len1, diag, gap_prob):
    L = wall_to_wall_distance(len0, len1, diag)
    expected_len = (2. / (2 - gap_prob)) * L
    assert expected_len >= 0
    return int(np.ceil(expected_len))


def band_radius(expected_len, gap_prob, sensitivity):
    assert 0 < gap_prob < 1 and 0 < sensitivity < 1
    epsilon = 1. - sensitivity
    C = erfcinv(epsilon) * np.sqrt(2 * gap_prob)
    radius = C * np.sqrt(expected_len)
    return max(1, int(np.ceil(radius)))


def band_radii(expected_lens, gap_prob, sensitivity):
    assert 0 < gap_prob < 1 and 0 < sensitivity < 1
    epsilon = 1. - sensitivity
    C = erfcinv(epsilon) * np.sqrt(2 * gap_prob)
    K = np.asarray(list(expected_lens), dtype=np.float64)
    return np.maximum(1, np.ceil(C * np.sqrt(K)).astype(np.int64))


def H0_moments(alphabet_len, wordlen, area):
    p_H0 = 1. / alphabet_len
    pw_H0 = p_H0 ** wordlen
    mu_H0 = area * pw_H0
    sd_H0 = np.sqrt(area * ((1 - pw_H0) * (pw_H0 + 2 * p_H0 * pw_H0 / (1 - p_H0)) - 2 * wordlen * pw_H0 ** 2))
    return mu_H0, sd_H0


def H1_moments(alphabet_len, wordlen, area, seglen, p_match):
    mu_H0, sd_H0 = H0_moments(alphabet_len, wordlen, area)
    p_H1 = p_match
    if p_H1 == 1.:
        p_H1 = 1 - np.finfo(float).eps
    pw_H1 = p_H1 ** wordlen
    mu_H1 = mu_H0 + seglen * pw_H1
    sd_H1 = np.sqrt(sd_H0 ** 2 + seglen * ((1 - pw_H1) * (pw_H1 + 2 * p_H1 * pw_H1 / (1 - p_H1)) - 2 * wordlen * pw_H1 ** 2))
    return mu_H1, sd_H1


class WordBlot(SeedIndex):
    """A similarity finder based on m-dependent CLT statistics (``blot.py:228-490``).

    Keyword Args:
        g_max (float): upper bound for indel probabilities.  sensitivity (float): desired band sensitivity.
        wordlen, alphabet, mask, device: as :class:`biseqt_amd.seeds.SeedIndex`.
    """

    def __init__(self, S, T, g_max=None, sensitivity=None, **kw):
        assert 0 < g_max < 1 and 0 < sensitivity < 1
        self.g_max = g_max
        self.sensitivity = sensitivity
        super(WordBlot, self).__init__(S, T, **kw)
        self._graph_key = None

    def _presentation(self):
        """Order in which the class iterates its seeds, as indices into the table rows (None = table order).  The
        in-memory *Ref variants scan T left to right instead (``blot.py:607-620``)."""
        return None

    def score_num_seeds(self, **kw):
        """z-scores of an observed number of seeds in a region against H0 and H1 (``blot.py:238-271``)."""
        num_seeds, area = kw['num_seeds'], kw['area']
        if area == 0:
            return float('-inf'), float('-inf')
        mu_H0, sd_H0 = H0_moments(len(self.alphabet), self.wordlen, area)
        mu_H1, sd_H1 = H1_moments(len(self.alphabet), self.wordlen, area, kw['seglen'], kw['p_match'])
        return (num_seeds - mu_H0) / sd_H0, (num_seeds - mu_H1) / sd_H1

    def band_radius(self, K):
        return band_radius(K, self.g_max, self.sensitivity)

    def segment_dims(self, d_band=None, a_band=None):
        """Edit path length and area of a diagonal / antidiagonal segment (``blot.py:283-303``; the reference
        divides with python 2's integer ``/``)."""
        a_min, a_max = a_band
        d_min, d_max = d_band
        K = (a_max - a_min) // 2
        A = (d_max - d_min) * K
        return K, A

    def estimate_match_probability(self, num_seeds, d_band=None, a_band=None):
        """``p = ((n - A p0^w) / K)^(1/w)``, 0 where the logarithm is undefined (``blot.py:305-341``)."""
        K, area = self.segment_dims(d_band=d_band, a_band=a_band)
        word_p_null = (1. / len(self.alphabet)) ** self.wordlen
        word_p = (num_seeds - area * word_p_null) / K
        if not word_p > 0:
            match_p = 0
        else:
            match_p = np.exp(np.log(word_p) / self.wordlen)
        return min(match_p, 1)

    def _graph(self, d_radius, a_radius):
        """The neighbourhood graph of ``find_all_neighbors`` (``blot.py:343-374``), built on the GPU once per
        (d_radius, a_radius)."""
        key = (d_radius, a_radius)
        if self._graph_key != key:
            self._idx.graph_build(1. * a_radius / d_radius, a_radius)
            self._graph_key = key

    def _seed_ps(self, K):
        d_radius = int(np.ceil(self.band_radius(K)))
        a_radius = K
        self._graph(d_radius, a_radius)
        n = self._idx.graph_counts().astype(np.int64)
        # every seed's segment has the same dimensions: K' = a_radius, A = 2 d_radius a_radius
        Kp, area = self.segment_dims(d_band=(-d_radius, d_radius), a_band=(-a_radius, a_radius))
        word_p_null = (1. / len(self.alphabet)) ** self.wordlen
        word_p = (n + 1 - area * word_p_null) / Kp
        p = np.zeros(len(n))
        pos = word_p > 0
        p[pos] = np.exp(np.log(word_p[pos]) / self.wordlen)
        return np.minimum(p, 1), d_radius, a_radius

    def score_seeds(self, K):
        """One dict per seed, in the class's seed order: ``seed`` (d, a), ``neighs`` (indices of the seeds in its
        neighbourhood), ``p`` estimated match probability of a segment centred there (``blot.py:376-408``)."""
        if not self.seed_count():
            return []
        p, _, _ = self._seed_ps(K)
        rows = self._idx.graph_points()
        if not len(rows):
            return []
        off, adj = self._idx.graph_fetch()
        perm = self._presentation()
        if perm is None:
            return [{'seed': (int(rows[k, 0]), int(rows[k, 1])), 'neighs': adj[off[k]:off[k + 1]].tolist(), 'p': p[k]}
                    for k in range(len(rows))]
        inv = np.empty(len(perm), np.int64)
        inv[perm] = np.arange(len(perm))
        return [{'seed': (int(rows[k, 0]), int(rows[k, 1])), 'neighs': inv[adj[off[k]:off[k + 1]]].tolist(), 'p': p[k]}
                for k in perm.tolist()]

    def similar_segments(self, K_min, p_min, at_least_one=False):
        """All maximal local similarities of a minimum length and match probability (``blot.py:410-490``): seeds
        with ``p >= p_min`` are grown into connected groups (the reference's depth-first search finds the
        connected components of the neighbourhood graph; they are computed on the GPU), each group's bounding
        segment is clamped to the table, scored and yielded in the order of its first seed."""
        if not self.seed_count():
            assert not at_least_one, 'no seeds found while at_least_one=True'
            return
        p, d_radius, a_radius = self._seed_ps(K_min)
        rows = self._idx.graph_points()              # the seeds as the class iterates them (mirrored for S == T)
        if not len(rows):
            assert not at_least_one, 'no seeds found while at_least_one=True'
            return
        perm = self._presentation()
        if perm is None:
            rank = np.arange(len(rows))                  # position of every table row in the class's seed order
        else:
            rank = np.empty(len(perm), np.int64)
            rank[perm] = np.arange(len(perm))
        avail = p >= p_min
        if not avail.any() and at_least_one:
            cand = np.flatnonzero(p == p.max())
            avail[cand[np.argmin(rank[cand])]] = True    # np.argmax over the class's list: its first maximum
        if not avail.any():
            return
        labels = self._idx.graph_components(avail)
        idx = np.flatnonzero(labels >= 0)
        order = idx[np.lexsort((rank[idx], labels[idx]))]       # grouped by component, seed order inside
        lab = labels[order]
        starts = np.flatnonzero(np.r_[True, lab[1:] != lab[:-1]])
        d, a = rows[order, 0].astype(np.int64), rows[order, 1].astype(np.int64)
        lenS, lenT = len(self.S), len(self.T)
        d_lo = np.minimum.reduceat(d, starts) - d_radius
        d_hi = np.maximum.reduceat(d, starts) + d_radius
        a_lo = np.minimum.reduceat(a, starts) - a_radius
        a_hi = np.maximum.reduceat(a, starts) + a_radius
        psum = np.add.reduceat(p[order], starts)
        cnt = np.diff(np.r_[starts, len(order)])
        firsts = order[starts]                                   # the seed each search starts from
        for s in np.argsort(rank[firsts], kind='stable').tolist():
            first = int(firsts[s])                       # ... and it is counted twice (:449,455)
            d_min = min(lenS, max(int(d_lo[s]), -lenT))
            d_max = min(lenS, max(int(d_hi[s]), -lenT))
            a_min = max(int(a_lo[s]), 0)
            a_max = min(int(a_hi[s]), lenS + lenT)
            seg = (d_min, d_max), (a_min, a_max)
            p_hat = (psum[s] + p[first]) / (cnt[s] + 1)
            res = {'segment': seg, 'p': p_hat}
            n = self.seed_count(d_band=seg[0], a_band=seg[1])
            K_hat, area_hat = self.segment_dims(d_band=seg[0], a_band=seg[1])
            res['scores'] = self.score_num_seeds(num_seeds=n, area=area_hat, seglen=K_hat, p_match=p_hat)
            yield res


class WordBlotOverlap(WordBlot):
    """Overlap (suffix-prefix) similarity detection between two sequences (``blot.py:228-236, 490-579``).

    Keyword Args:
        g_max (float): upper bound for indel probabilities.  sensitivity (float): desired band sensitivity.
        wordlen, alphabet, mask, device: as :class:`biseqt_amd.seeds.SeedIndex`.
    """

    def __init__(self, S, T, g_max=None, sensitivity=None, **kw):
        super(WordBlotOverlap, self).__init__(S, T, g_max=g_max, sensitivity=sensitivity, **kw)
        assert not self.self_comp, 'overlap detection compares two different sequences'

    def _tables(self):
        """L(d) and r(d) for every diagonal d = -|T| .. |S| (index d + |T|), the reference's `_len` / `_rad`."""
        lenS, lenT = len(self.S), len(self.T)
        d = np.arange(-lenT, lenS + 1, dtype=np.int64)
        wall = np.minimum(lenS - d, lenT) + np.minimum(d, 0)
        L = np.ceil((2. / (2 - self.g_max)) * wall).astype(np.int64)
        rad = band_radii(L, self.g_max, self.sensitivity)
        return L, rad

    def score_seeds(self):
        """One dict per seed, in table order: ``seed`` (d, a), ``r`` band radius at its diagonal, ``L`` expected
        overlap length, ``p`` estimated match probability (``blot.py:497-556``)."""
        rows = self.rows()
        if not len(rows):
            return []
        L_tab, r_tab = self._tables()
        n = self._idx.band_neighbours(r_tab.astype(np.float64)).astype(np.int64)
        lenT = len(self.T)
        d = rows[:, 0].astype(np.int64)
        L = L_tab[d + lenT]
        rad = r_tab[d + lenT]
        area = 2 * rad * L
        word_p_null = (1. / len(self.alphabet)) ** self.wordlen
        word_p = (n + 1 - area * word_p_null) / L
        pos = word_p > 0                      # log(x <= 0) warns, which the reference answers with p = 0 (:541-545)
        p = np.zeros(len(rows))
        p[pos] = np.exp(np.log(word_p[pos]) / self.wordlen)
        p = np.minimum(p, 1)
        perm = self._presentation()
        ks = range(len(rows)) if perm is None else perm.tolist()
        return [{'seed': (int(rows[k, 0]), int(rows[k, 1])), 'r': np.float64(rad[k]), 'L': int(L[k]), 'p': p[k]}
                for k in ks]

    def highest_scoring_overlap_band(self):
        """The diagonal band with the highest estimated match probability: ``d_band``, ``p``, ``len``, ``score``
        (z-score under H1), or None without seeds (``blot.py:558-579``)."""
        scored = self.score_seeds()
        if not scored:
            return None
        idx = max(range(len(scored)), key=lambda i: scored[i]['p'])
        seed, rad = scored[idx]['seed'], scored[idx]['r']
        p_hat, overlap_len = scored[idx]['p'], scored[idx]['L']
        d_band = seed[0] - rad, seed[0] + rad
        res = {'d_band': d_band, 'p': p_hat, 'len': overlap_len}
        area = 2 * rad * overlap_len
        mu_H1, sd_H1 = H1_moments(len(self.alphabet), self.wordlen, area, overlap_len, p_hat)
        num_seeds = self.seed_count(d_band=d_band)
        res['score'] = (num_seeds - mu_H1) / sd_H1
        return res


def _check_ref_memory(alphabet, wordlen, allowed_memory):
    """The in-memory variants refuse word lengths whose k-mer table would not fit the allowed memory
    (``blot.py:596-604``: one python int -- 24 bytes under python 2 -- per possible k-mer)."""
    num_kmers = len(alphabet) ** wordlen
    mem_needed_gb = np.power(2, np.log2(24. * num_kmers) - 30)
    assert allowed_memory > 0, 'allowed memory must be positive'
    if mem_needed_gb > allowed_memory:
        raise MemoryError('not enough memory (max = %.2f GB) to store %d-mers (%.2f GB needed)'
                          % (allowed_memory, wordlen, mem_needed_gb))


class _RefMixin(object):
    """Shared behaviour of the reference's in-memory classes: built around ONE sequence (``ref``), every query
    names the other one, and seeds are iterated with T scanned left to right, hits of S ascending
    (``blot.py:607-620``) -- i.e. the table rows ordered by (j, i)."""

    def _init_ref(self, ref, allowed_memory, kw):
        self._ref_kw = dict(kw)
        self.allowed_memory = allowed_memory
        _check_ref_memory(kw['alphabet'], kw['wordlen'], allowed_memory)
        self.wordlen, self.alphabet = kw['wordlen'], kw['alphabet']
        self.g_max, self.sensitivity = kw['g_max'], kw['sensitivity']
        self.S, self.T = ref, None
        self._idx = None

    def _set_query(self, seq):
        if self.T is not None and self.T == seq and self._idx is not None:
            return
        if self._idx is not None:
            self._idx.close()
        kw = dict(self._ref_kw)
        g_max, sensitivity = kw.pop('g_max'), kw.pop('sensitivity')
        self._base.__init__(self, self.S, seq, g_max=g_max, sensitivity=sensitivity, **kw)
        self._perm = None

    def _points(self):
        """The (d, a) points the class works on, in the order the device lists them: the table rows, or -- when the
        query IS the reference (``blot.py:612``: trivial seeds skipped, every other pair met from both sides) -- each
        non-trivial row followed by its mirror image."""
        rows = self.rows().astype(np.int64)
        if not self.self_comp:
            return rows
        nt = rows[rows[:, 0] != 0]
        pts = np.empty((2 * len(nt), 2), np.int64)
        pts[0::2] = nt
        pts[1::2] = nt * np.array([-1, 1])
        return pts

    def _presentation(self):
        if self._perm is None:
            pts = self._points()
            i, j = (pts[:, 1] + pts[:, 0]) // 2, (pts[:, 1] - pts[:, 0]) // 2
            self._perm = np.lexsort((i, j))
        return self._perm

    def seeds(self, exclude_trivial=True):
        assert self.T is not None
        pts = self._points()[self._presentation()]
        return list(zip(((pts[:, 1] + pts[:, 0]) // 2).tolist(), ((pts[:, 1] - pts[:, 0]) // 2).tolist()))

    def seed_count(self, d_band=None, a_band=None):
        """The in-memory classes count their own seed list (``blot.py:622-637, 683-698``), not table rows: for a self
        comparison that is every non-trivial row and its mirror image."""
        if not self.self_comp:
            return self._base.seed_count(self, d_band=d_band, a_band=a_band)

        def nontrivial(db):
            c = self._idx.count(db, a_band)
            if db is None or db[0] <= 0 <= db[1]:
                c -= self._idx.count((0, 0), a_band)
            return c
        mirror = None if d_band is None else (-d_band[1], -d_band[0])
        return nontrivial(d_band) + nontrivial(mirror)


class WordBlotLocalRef(_RefMixin, WordBlot):
    """In-memory variant of :class:`WordBlot` (``blot.py:626-700``): ``WordBlotLocalRef(ref, allowed_memory=1, **kw)``,
    then ``similar_segments(seq, K_min, p_min)`` / ``score_seeds_(seq, K)`` for any number of query sequences."""
    _base = WordBlot

    def __init__(self, ref, allowed_memory=1, **kw):
        self._init_ref(ref, allowed_memory, kw)

    def score_seeds_(self, seq, K):
        self._set_query(seq)
        return WordBlot.score_seeds(self, K)

    def similar_segments(self, seq, K_min, p_min, at_least_one=False):
        self._set_query(seq)
        return WordBlot.similar_segments(self, K_min, p_min, at_least_one=at_least_one)


class WordBlotOverlapRef(_RefMixin, WordBlotOverlap):
    """In-memory variant of :class:`WordBlotOverlap` (``blot.py:582-624``)."""
    _base = WordBlotOverlap

    def __init__(self, ref, allowed_memory=1, **kw):
        self._init_ref(ref, allowed_memory, kw)

    def score_seeds_(self, seq):
        self._set_query(seq)
        return WordBlotOverlap.score_seeds(self)

    def highest_scoring_overlap_band(self, seq):
        self._set_query(seq)
        return WordBlotOverlap.highest_scoring_overlap_band(self)
