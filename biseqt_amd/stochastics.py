"""Random sequences, the mutation process and its log-odds scores.

Mirrors the reference's ``biseqt/stochastics.py``: ``rand_seq`` (:28-41), ``rand_read`` (:44-88),
``MutationProcess.mutate`` (:143-201), ``MutationProcess.noisy_read`` (:203-232) and
``MutationProcess.log_odds_scores`` (:234-310).  The log-odds formula defines the floating-point
scoring case of the aligner.  Randomness comes from a ``numpy.random.Generator`` that callers may
pass in (``rng=``); the module-level default is seeded from OS entropy like the reference's use of
the global numpy RNG.
"""
from itertools import product
from math import log

import numpy as np

from .sequence import Alphabet, Sequence

_default_rng = np.random.default_rng()


def rand_seq(alphabet, size, p=None, rng=None):
    """A random :class:`Sequence` of the given length; letters i.i.d. with distribution ``p``
    (uniform by default)."""
    assert isinstance(alphabet, Alphabet)
    rng = rng or _default_rng
    contents = rng.choice(len(alphabet), size=int(size), p=p)
    return Sequence(alphabet, contents.tolist())


def rand_read(seq, len_mean=None, len_sd=1, expected_coverage=None, num=None, rng=None):
    """Lossless reads of ``seq``: substrings of Gaussian length at uniformly chosen starts; ``num`` reads, or as
    many as ``expected_coverage`` asks for, or one (``stochastics.py:44-88``).  Yields ``(read, start)``."""
    assert len_mean < len(seq), 'Expected read length must be smaller than the sequence length'
    assert num is None or expected_coverage is None, 'At most one of expected_coverage or num can be specified'
    rng = rng or _default_rng
    if num is None:
        num = 1 if expected_coverage is None else int(1. * len(seq) * expected_coverage / len_mean)
    for length in rng.normal(loc=len_mean, scale=len_sd, size=num):
        length = max(1, min(len(seq) - 1, int(length)))     # at least 1, at most |S| - 1
        start = int(rng.choice(len(seq) - length))
        yield seq[start:start + length], start


class MutationProcess(object):
    """Substitutions + affine-length indels (reference ``stochastics.py:91-141``).

    ``subst_probs`` is either a full L x L row-stochastic matrix or a single number: the probability
    of *any* substitution, spread evenly over the other letters.  A gap opens with probability
    ``go_prob`` after a non-gap operation (insertion or deletion with equal chance) and is extended
    with probability ``ge_prob``; ``go_prob <= ge_prob`` is required.
    """

    def __init__(self, alphabet, subst_probs=None, ge_prob=0, go_prob=0, insert_dist=None, rng=None):
        assert isinstance(alphabet, Alphabet)
        self.alphabet = alphabet
        self.rng = rng or _default_rng
        if not isinstance(subst_probs, list):
            L = len(self.alphabet)
            assert subst_probs < 1 and subst_probs >= 0
            any_subst = float(subst_probs)
            each_subst = any_subst / (L - 1)
            match = 1 - any_subst
            subst_probs = [[match if i == j else each_subst for j in range(L)] for i in range(L)]
        self.subst_probs = subst_probs
        assert go_prob < 1 and ge_prob < 1 and go_prob >= 0 and ge_prob >= 0
        assert go_prob <= ge_prob, \
            'Gap-open probability cannot be larger than gap-extend probability'
        self.go_prob, self.ge_prob = go_prob, ge_prob
        self.insert_dist = insert_dist

    def mutate(self, seq):
        """Copy ``seq`` position by position, substituting, opening and extending gaps at random;
        returns the mutant and the edit transcript that produced it (``stochastics.py:143-201``)."""
        L = len(self.alphabet)
        rng = self.rng
        pos, T, op, opseq = 0, [], '', []
        while pos < len(seq):
            if op and op in 'ID':
                if rng.random() < self.ge_prob:          # extend the open gap
                    if op == 'I':
                        T.append(int(rng.choice(L, p=self.insert_dist)))
                    else:
                        pos += 1
                else:
                    op = ''                                # the gap ends; nothing is emitted
            else:
                if rng.random() < self.go_prob:          # open a gap, either kind with equal chance
                    if rng.random() < 0.5:
                        op = 'D'
                        pos += 1
                    else:
                        op = 'I'
                        T.append(int(rng.choice(L, p=self.insert_dist)))
                else:
                    copy = int(rng.choice(L, p=self.subst_probs[seq[pos]]))
                    T.append(copy)
                    op = 'M' if copy == seq[pos] else 'S'
                    pos += 1
            opseq.append(op)
        return Sequence(self.alphabet, T), ''.join(opseq)

    def noisy_read(self, seq, **kw):
        """:func:`rand_read`, then every read through :func:`mutate` (``stochastics.py:203-232``).  Yields
        ``(noisy read, start of the lossless read, edit transcript)``."""
        kw.setdefault('rng', self.rng)
        for read, start in rand_read(seq, **kw):
            read, tx = self.mutate(read)
            yield read, start, tx

    def log_odds_scores(self, null_hypothesis=None):
        """Natural-log odds scores of the process (``stochastics.py:234-310``):

        ``S(a_i -> a_j) = log(1 - g_e) + log Pr(a_j | a_i) - log Pr_0(a_j)`` and the affine gap scores
        ``(log g_o - log g_e, log g_e)``.  Returns ``(subst_scores, (go_score, ge_score))``.
        """
        L = len(self.alphabet)
        if null_hypothesis is None:
            null_hypothesis = [1. / L] * L
        err = 'Zero probabilities are not allowed for score calculation'
        assert all(x > 0 and x < 1 for x in null_hypothesis), err
        assert all(x > 0 and x < 1 for y in self.subst_probs for x in y), err
        assert self.ge_prob > 0 and self.go_prob > 0, err
        subst_scores = [[0] * L for _ in range(L)]
        for i, j in product(range(L), repeat=2):
            subst_scores[i][j] = log(1 - self.ge_prob) + \
                log(self.subst_probs[i][j]) - log(null_hypothesis[j])
        gap_scores = log(self.go_prob) - log(self.ge_prob), log(self.ge_prob)
        return subst_scores, gap_scores
