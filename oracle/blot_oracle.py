"""CPU restatement of the reference's overlap band selection (TEST INFRASTRUCTURE, never the product path).

Follows /root/reference/biseqt/blot.py:
  * find_peaks                      blot.py:37-76
  * wall_to_wall_distance           blot.py:78-89
  * expected_overlap_len            blot.py:92-112
  * band_radius / band_radii        blot.py:116-160
  * H0_moments / H1_moments         blot.py:163-218
  * score_seeds                     blot.py:497-556   (WordBlotOverlap; the neighbour search is scipy's cKDTree,
                                                      exactly as the reference calls it)
  * highest_scoring_overlap_band    blot.py:558-579
The module-level `warnings.filterwarnings('error')` of the reference (blot.py:33-34) turns numpy's warnings for
log(x <= 0) into exceptions, which `_p` answers with match_p = 0 (blot.py:541-545); restated as a branch.

Parity pin: the reference's Python cannot be imported here; pinned by the closed-form properties of
the reference's tests/test_blot.py:35-114, the two values SURVEY 8f records from the reference
(band_radius(2000, .2, .99) == 52, expected_overlap_len(5000, 5000, 1000, .2) == 4445) and its statistical
overlap-detection test (:160-197), restated in tests/test_blot_oracle.py.
"""
import numpy as np
from scipy.spatial import cKDTree
from scipy.special import erfcinv

from . import seeds_oracle as SO


def find_peaks(xs, rs, threshold):
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
    return np.array([max(1, int(np.ceil(C * np.sqrt(K)))) for K in expected_lens])


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


def match_p(n, d_radius, L, alphabet_len, wordlen):
    """`_p` of blot.py:533-548 for a seed with n neighbours (itself excluded)."""
    area = 2 * d_radius * L
    word_p_null = (1. / alphabet_len) ** wordlen
    word_p = (n + 1 - area * word_p_null) / L
    if not word_p > 0:                     # np.log warns -> the reference's `except Warning` branch
        p = 0
    else:
        p = np.exp(np.log(word_p) / wordlen)
    return min(p, 1)


def _seed_list(S, T, wordlen, alphabet_len, mask, order):
    """(i, j) seeds in the order the class iterates them: 'table' = SeedIndex.seeds(exclude_trivial=True)
    (seeds.py:164-197), 'mutant' = the in-memory *Ref classes (blot.py:607-620: T scanned left to right)."""
    if order == 'mutant':
        assert not mask
        return SO.seeds_by_mutant(S, T, wordlen, alphabet_len, exclude_trivial=True)
    rows, self_comp = SO.seed_rows(S, T, wordlen, alphabet_len, mask)
    return SO.seeds(rows, self_comp, exclude_trivial=True)


def _seed_count(S, T, wordlen, alphabet_len, mask, order, d_band=None, a_band=None):
    """`seed_count` as the class at hand counts: the SQL classes count rows of the seeds table (seeds.py:199-231), the
    in-memory *Ref classes iterate their own `seeds()` list, trivial seeds excluded (blot.py:622-637, 683-698)."""
    if order != 'mutant':
        rows, _ = SO.seed_rows(S, T, wordlen, alphabet_len, mask)
        return SO.seed_count(rows, d_band=d_band, a_band=a_band)
    cnt = 0
    for i, j in _seed_list(S, T, wordlen, alphabet_len, mask, order):
        d, a = SO.to_diagonal_coordinates(i, j)
        if d_band and not d_band[0] <= d <= d_band[1]:
            continue
        if a_band and not a_band[0] <= a <= a_band[1]:
            continue
        cnt += 1
    return cnt


def score_seeds(S, T, wordlen, alphabet_len, g_max, sensitivity, mask=(), order='table'):
    ij = _seed_list(S, T, wordlen, alphabet_len, mask, order)
    all_seeds = [SO.to_diagonal_coordinates(i, j) for i, j in ij]
    if not all_seeds:
        return []
    lenS, lenT = len(S), len(T)

    def _len(d):
        return expected_overlap_len(lenS, lenT, d, g_max)

    def _rad(d):
        return np.ceil(band_radius(_len(d), g_max, sensitivity))

    scaled = np.array([(d / _rad(d),) for d, a in all_seeds])
    tree = cKDTree(scaled)
    neighs = tree.query_ball_tree(tree, 1, p=float('inf'))
    out = []
    for idx, (d, a) in enumerate(all_seeds):
        n = len(neighs[idx]) - 1
        L = _len(d)
        out.append({'seed': (d, a), 'r': _rad(d), 'L': L,
                    'p': match_p(n, int(np.ceil(band_radius(L, g_max, sensitivity))), L, alphabet_len, wordlen)})
    return out


def highest_scoring_overlap_band(S, T, wordlen, alphabet_len, g_max, sensitivity, mask=(), order='table'):
    scored = score_seeds(S, T, wordlen, alphabet_len, g_max, sensitivity, mask, order)
    if not scored:
        return None
    idx = max(range(len(scored)), key=lambda i: scored[i]['p'])
    seed, rad = scored[idx]['seed'], scored[idx]['r']
    p_hat, overlap_len = scored[idx]['p'], scored[idx]['L']
    d_band = seed[0] - rad, seed[0] + rad
    res = {'d_band': d_band, 'p': p_hat, 'len': overlap_len}
    area = 2 * rad * overlap_len
    mu_H1, sd_H1 = H1_moments(alphabet_len, wordlen, area, overlap_len, p_hat)
    num_seeds = _seed_count(S, T, wordlen, alphabet_len, mask, order, d_band=d_band)
    res['score'] = (num_seeds - mu_H1) / sd_H1
    return res


# ---- local similarities: WordBlot.score_seeds / similar_segments (blot.py:283-490) --------------------------------
def segment_dims(d_band, a_band):                       # blot.py:283-303 (python-2 integer division)
    a_min, a_max = a_band
    d_min, d_max = d_band
    K = (a_max - a_min) // 2
    A = (d_max - d_min) * K
    return K, A


def estimate_match_probability(num_seeds, d_band, a_band, alphabet_len, wordlen):     # blot.py:305-341
    K, area = segment_dims(d_band, a_band)
    word_p_null = (1. / alphabet_len) ** wordlen
    word_p = (num_seeds - area * word_p_null) / K
    if not word_p > 0:
        p = 0
    else:
        p = np.exp(np.log(word_p) / wordlen)
    return min(p, 1)


def score_num_seeds(num_seeds, area, seglen, p_match, alphabet_len, wordlen):         # blot.py:238-271
    if area == 0:
        return float('-inf'), float('-inf')
    mu_H0, sd_H0 = H0_moments(alphabet_len, wordlen, area)
    mu_H1, sd_H1 = H1_moments(alphabet_len, wordlen, area, seglen, p_match)
    return (num_seeds - mu_H0) / sd_H0, (num_seeds - mu_H1) / sd_H1


def find_all_neighbors(all_seeds, d_radius, a_radius):                                # blot.py:343-374
    d_coeff = 1. * a_radius / d_radius
    if not all_seeds:
        return []
    scaled = np.array([(d * d_coeff, a) for d, a in all_seeds])
    tree = cKDTree(scaled)
    neighs = tree.query_ball_tree(tree, a_radius, p=float('inf'))
    for idx, _ in enumerate(neighs):
        neighs[idx].remove(idx)
    return list(zip(all_seeds, neighs))


def score_seeds_local(S, T, wordlen, alphabet_len, g_max, sensitivity, K, mask=(), order='table'):   # blot.py:376-408
    ij = _seed_list(S, T, wordlen, alphabet_len, mask, order)
    all_seeds = [SO.to_diagonal_coordinates(i, j) for i, j in ij]
    d_radius = int(np.ceil(band_radius(K, g_max, sensitivity)))
    a_radius = K
    out = []
    for (d, a), neighs in find_all_neighbors(all_seeds, d_radius, a_radius):
        p = estimate_match_probability(len(neighs) + 1, (d - d_radius, d + d_radius), (a - a_radius, a + a_radius),
                                       alphabet_len, wordlen)
        out.append({'seed': (d, a), 'neighs': neighs, 'p': p})
    return out


def similar_segments(S, T, wordlen, alphabet_len, g_max, sensitivity, K_min, p_min, at_least_one=False, mask=(),
                     order='table'):
    """blot.py:410-490, the depth-first growth included (the order of `ps_in_seg`, and with it the last bits of the
    averaged p, follows the KD-tree's neighbour order)."""
    d_radius = int(np.ceil(band_radius(K_min, g_max, sensitivity)))
    a_radius = K_min
    scored = score_seeds_local(S, T, wordlen, alphabet_len, g_max, sensitivity, K_min, mask, order)
    lenS, lenT = len(S), len(T)
    avail = [rec['p'] >= p_min for rec in scored]
    if not any(avail) and at_least_one:
        assert len(scored)
        avail[int(np.argmax([rec['p'] for rec in scored]))] = True
    out = []
    while True:
        try:
            seed_idx = avail.index(True)
        except ValueError:
            break
        stack = [seed_idx]
        avail[seed_idx] = False
        ps = [scored[seed_idx]['p']]
        seg = None
        while stack:
            idx = stack.pop()
            ps.append(scored[idx]['p'])
            d, a = scored[idx]['seed']
            if seg is None:
                seg = (d - d_radius, d + d_radius), (a - a_radius, a + a_radius)
            else:
                (d_min, d_max), (a_min, a_max) = seg
                seg = (min(d - d_radius, d_min), max(d + d_radius, d_max)), (min(a - a_radius, a_min), max(a + a_radius, a_max))
            for neigh in scored[idx]['neighs']:
                if avail[neigh]:
                    stack.append(neigh)
                    avail[neigh] = False
        (d_min, d_max), (a_min, a_max) = seg
        d_min = min(lenS, max(d_min, -lenT)); d_max = min(lenS, max(d_max, -lenT))
        a_min = max(a_min, 0); a_max = min(a_max, lenS + lenT)
        seg = (d_min, d_max), (a_min, a_max)
        p_hat = sum(ps) / len(ps)
        n = _seed_count(S, T, wordlen, alphabet_len, mask, order, d_band=seg[0], a_band=seg[1])
        K_hat, area_hat = segment_dims(seg[0], seg[1])
        out.append({'segment': seg, 'p': p_hat,
                    'scores': score_num_seeds(n, area_hat, K_hat, p_hat, alphabet_len, wordlen)})
    return out
