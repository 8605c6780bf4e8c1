"""k-mers as integers (host side of include/pw_seeds.h).

Mirrors ``biseqt/kmers.py``: :func:`kmer_as_int` (:164-210) and :func:`as_kmer_seq` (:213-241) with the same
arguments and return values.  ``kmer_as_int`` is plain integer arithmetic on a handful of letters and runs on
the host; ``as_kmer_seq`` of a whole sequence is computed by the GPU encoder (kernel K5a).  The SQLite-backed
``KmerIndex`` / ``KmerCache`` of the reference are storage, not computation, and have no counterpart: the
sorted (k-mer, position) arrays live in HBM inside :class:`biseqt_amd.seeds.SeedIndex`.
"""
from .sequence import Alphabet, Sequence

DIGITS = '0123456789abcdefghijklmnopqrstuvwxyz'     # kmers.py:162: at most 36 letters


def check_limits(alphabet, wordlen):
    """The reference's limits (``kmers.py:262-271``): |alphabet| <= 36, wordlen < 31.5 on 64-bit."""
    assert isinstance(wordlen, int)
    assert isinstance(alphabet, Alphabet)
    assert len(alphabet) <= len(DIGITS), 'Maximum alphabet size of %d exceeded' % len(DIGITS)
    assert wordlen < (64 - 1) / 2., 'Maximum kmer length %d for %d-bit integers exceeded' % (wordlen, 64)
    assert len(alphabet) ** wordlen < 2 ** 62, 'alphabet size ** word length must stay below 2^62'


def kmer_as_int(contents, alphabet):
    """Integer representation of a k-mer: its letters as digits in base ``len(alphabet)``."""
    assert isinstance(alphabet, Alphabet)
    v, L = 0, len(alphabet)
    for c in contents:
        v = v * L + int(c)
    return v


def mask_bits(mask):
    """A list of sets of letter indices -> the bit masks the C ABI takes."""
    assert all(isinstance(lets, set) for lets in mask)
    bits = []
    for lets in mask:
        b = 0
        for c in lets:
            b |= 1 << int(c)
        bits.append(b)
    return bits


def as_kmer_seq(seq, wordlen, mask=[]):
    """The k-mer at every position of ``seq`` as an integer, ``None`` where the set of letters of the k-mer
    equals one of the sets in ``mask`` (computed on the GPU)."""
    assert isinstance(seq, Sequence)
    from .seeds import _Index
    with _Index(seq, seq, wordlen, seq.alphabet, mask, self_comp=1) as idx:
        ks = idx.kmers(0)
    return [None if k < 0 else int(k) for k in ks]
