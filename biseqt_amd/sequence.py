"""Alphabets and sequences: the minimal host-side data model the alignment API needs.

Mirrors the interface of the reference's ``biseqt/sequence.py`` (``Alphabet`` :31-161, ``Sequence``
:164-234) so code written against ``biseqt.sequence`` keeps working: a sequence is an immutable tuple
of letter indices into an alphabet whose letters all have the same printed length.
"""
from hashlib import sha1
from itertools import chain

import numpy as np


class Alphabet(object):
    """An ordered set of equally long letters (reference ``sequence.py:31-57``)."""

    def __init__(self, letters):
        self._letters = tuple(letters)
        self._letlen = len(self._letters[0])
        assert all(len(l) == self._letlen for l in self._letters), \
            'All alphabet letters must have the same length'
        self._idx_by_letter = {l: idx for idx, l in enumerate(self._letters)}

    def letter_to_idx(self, letters):
        """Letters -> tuple of their positions in the alphabet (``sequence.py:58-71``)."""
        return tuple(self._idx_by_letter[l] for l in letters)

    def parse(self, string):
        """String -> :class:`Sequence`, cutting it into letters of ``_letlen`` characters
        (``sequence.py:73-91``)."""
        assert isinstance(string, str), 'Raw sequence must be in string form'
        assert len(string) % self._letlen == 0, 'String representation ' + \
            'of sequence must be a multiple of the alphabet letter length'
        n = self._letlen
        pieces = [string[k:k + n] for k in range(0, len(string), n)]
        return Sequence(self, self.letter_to_idx(pieces))

    def transform(self, seq, mappings={}):
        """Letter-to-letter translation of a sequence; ``mappings`` is a dict of rules or a list of
        two-element bidirectional rules, entries being letters or indices (``sequence.py:93-141``)."""
        mappings = mappings if mappings is not None else {}
        if isinstance(mappings, list):
            assert all(len(m) == 2 for m in mappings)
            mappings = dict(chain.from_iterable(
                [(rule[0], rule[1]), (rule[1], rule[0])] for rule in mappings))
        pair_of = list(range(len(self)))
        for key, val in mappings.items():
            if not isinstance(key, int):
                key = self._idx_by_letter[key]
            if not isinstance(val, int):
                val = self._idx_by_letter[val]
            pair_of[key] = val
        return Sequence(self, tuple(pair_of[c] for c in seq))

    def __len__(self):
        return len(self._letters)

    def __eq__(self, other):
        assert isinstance(other, Alphabet), 'Only alphabets can be compared with alphabets'
        return self._letters == other._letters

    def __ne__(self, other):
        return not self.__eq__(other)

    def __hash__(self):
        return hash(self._letters)

    def __getitem__(self, key):
        return self._letters.__getitem__(key)

    def __repr__(self):
        return 'Alphabet([%s])' % ','.join('"%s"' % self[idx] for idx in range(len(self)))


class Sequence(object):
    """An immutable sequence of letter indices (reference ``sequence.py:164-234``)."""

    def __init__(self, alphabet, contents=()):
        assert isinstance(alphabet, Alphabet)
        self.alphabet = alphabet
        if isinstance(contents, np.ndarray):
            assert contents.ndim == 1 and (contents.dtype.kind in 'iu' or contents.size == 0)
            assert contents.size == 0 or (0 <= int(contents.min()) and int(contents.max()) < len(alphabet))
            contents = tuple(contents.tolist())
        elif isinstance(contents, Sequence):
            contents = contents.contents
        else:
            contents = tuple(int(c) if isinstance(c, (np.integer,)) else c for c in contents)
            assert all(isinstance(c, int) and 0 <= c < len(alphabet) for c in contents)
        self.contents = contents

    @property
    def content_id(self):
        """Hex SHA-1 of the printed sequence (reference ``sequence.py:190`` computes it in the constructor; here it is
        computed on first use -- sequences of megabases are sliced often and compared rarely)."""
        cid = self.__dict__.get('_content_id')
        if cid is None:
            cid = sha1(str(self).encode('utf-8')).hexdigest()
            self.__dict__['_content_id'] = cid
        return cid

    def as_array(self, dtype=np.uint8):
        """The contents as a numpy array (one byte per letter is what the device arena holds)."""
        cache = self.__dict__.setdefault('_arrays', {})           # contents are immutable
        key = np.dtype(dtype).str
        if key not in cache:
            arr = np.asarray(self.contents, dtype=dtype)
            arr.setflags(write=False)
            cache[key] = arr
        return cache[key]

    def reverse(self):
        return Sequence(self.alphabet, tuple(reversed(self.contents)))

    def transform(self, mappings={}):
        return self.alphabet.transform(self, mappings=mappings)

    def __str__(self):
        return ''.join(self.alphabet[idx] for idx in self.contents)

    def __repr__(self):
        return 'Sequence(%s, contents=%s)' % (repr(self.alphabet), repr(self.contents))

    def __len__(self):
        return len(self.contents)

    def __bool__(self):
        return bool(self.contents)

    __nonzero__ = __bool__

    def __iter__(self):
        return iter(self.contents)

    def __getitem__(self, key):
        if isinstance(key, (int, np.integer)):
            return self.contents[key]
        sub = Sequence.__new__(Sequence)                 # a slice of valid contents needs no re-validation
        sub.alphabet = self.alphabet
        sub.contents = self.contents.__getitem__(key)
        return sub

    def __eq__(self, other):
        # the reference compares content ids (sequence.py:224-226); equal alphabets print equal contents equally
        return self.alphabet == other.alphabet and self.contents == other.contents

    def __ne__(self, other):
        return not self.__eq__(other)

    def __hash__(self):
        return hash((self.alphabet, self.content_id))

    def __add__(self, other):
        if isinstance(other, Sequence):
            assert self.alphabet == other.alphabet
            contents = other.contents
        else:
            contents = self.alphabet.letter_to_idx(other)
        return Sequence(self.alphabet, self.contents + contents)
