"""Synthetic workloads for tests and bench.py (this build's own generator).

NumPy ``Generator(PCG64(seed))``, uniform letters as in the reference's ``rand_seq``
(reference ``biseqt/stochastics.py:28-41``); mutants come from a vectorised edit process with the
same three knobs as the reference's ``MutationProcess`` (``stochastics.py:143-201``): per-position
substitution probability, gap-open probability (insertion or deletion with equal chance) and
geometric gap extension.  It is *a* generator of realistic inputs, not a re-implementation of that
Markov chain draw for draw.
"""
import numpy as np


def rng_for(seed):
    return np.random.Generator(np.random.PCG64(seed))


def rand_seqs(rng, n_seqs, length, L=4):
    """``n_seqs`` uniform random sequences over an ``L``-letter alphabet, uint8 [n_seqs, length]."""
    return rng.integers(0, L, size=(n_seqs, length), dtype=np.uint8)


def mutate(rng, origin, subst=0.05, go=0.03, ge=0.03, L=4):
    """Mutate one uint8 sequence; returns the mutant (uint8, ragged length)."""
    origin = np.asarray(origin, dtype=np.uint8)
    n = len(origin)
    if n == 0:
        return origin.copy()
    opens = rng.random(n) < go
    is_del = rng.random(n) < 0.5
    glen = rng.geometric(1.0 - ge, size=n)            # >= 1
    # deletions: a run of glen positions starting at an opening position
    delta = np.zeros(n + 1, dtype=np.int64)
    dstart = np.nonzero(opens & is_del)[0]
    dend = np.minimum(dstart + glen[dstart], n)
    np.add.at(delta, dstart, 1)
    np.add.at(delta, dend, -1)
    deleted = np.cumsum(delta[:n]) > 0
    # insertions: glen random letters in front of an opening position
    ins = np.where(opens & ~is_del, glen, 0)
    keep = (~deleted).astype(np.int64)
    # substitutions on kept letters
    letters = origin.copy()
    sub = rng.random(n) < subst
    shift = rng.integers(1, max(L, 2), size=n, dtype=np.uint8)
    letters = np.where(sub, (letters + shift) % L, letters).astype(np.uint8)
    total = int(ins.sum() + keep.sum())
    out = np.empty(total, dtype=np.uint8)
    # position i contributes ins[i] random letters then (keep[i]) its own letter
    counts = ins + keep
    ends = np.cumsum(counts)
    own_pos = ends - 1                                  # slot of the kept letter (if kept)
    is_own = np.zeros(total, dtype=bool)
    is_own[own_pos[keep == 1]] = True
    out[is_own] = letters[keep == 1]
    out[~is_own] = rng.integers(0, L, size=int((~is_own).sum()), dtype=np.uint8)
    return out


def pair_batch(seed, n_pairs, length, subst=0.05, go=0.03, ge=0.03, L=4):
    """BASELINE config-2 style batch: ``n_pairs`` (origin, mutant) with |origin| = ``length``.

    Returns (origins, mutants): lists of uint8 arrays.
    """
    rng = rng_for(seed)
    origins = rand_seqs(rng, n_pairs, length, L)
    mutants = [mutate(rng, origins[k], subst, go, ge, L) for k in range(n_pairs)]
    return [origins[k] for k in range(n_pairs)], mutants


def banded_cells(X, Y, dmin, dmax):
    """Cells the reference allocates for a banded table (reference ``_pw_internals.c:29-36,53-56``),
    the denominator of the GCUPS metric."""
    dmax = min(dmax, X)
    dmin = max(dmin, -Y)
    if dmax < dmin:
        return 0
    d = np.arange(dmin, dmax + 1, dtype=np.int64)
    return int((1 + np.minimum(d, 0) + np.minimum(X - d, Y)).sum())


def std_cells(X, Y):
    return (X + 1) * (Y + 1)
