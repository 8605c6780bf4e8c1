"""Shared test helpers: golden-vector loading and result comparison."""
import gzip
import hashlib
import json
import os

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_golden(name):
    path = os.path.join(GOLDEN, name)
    if name.endswith('.gz'):
        with gzip.open(path) as f:
            return json.loads(f.read().decode())['records']
    with open(path) as f:
        return json.load(f)['records']


def dec(s):
    return [ord(c) - ord('0') for c in s]


def kw_of(rec):
    kw = dict(rec['kw'])
    for k in ('origin_range', 'mutant_range', 'diag_range'):
        if k in kw:
            kw[k] = tuple(kw[k])
    kw.setdefault('go', 0.)
    kw.setdefault('ge', 0.)
    if kw.get('subst') is None:
        kw.setdefault('match', 1.)       # the reference's defaults (pw.py:185-195)
        kw.setdefault('mismatch', 0.)
    if 'kw_hex' in rec:      # exact float scores
        kw['subst'] = [[float.fromhex(v) for v in row] for row in rec['kw_hex']['subst']]
        kw['go'] = float.fromhex(rec['kw_hex']['go'])
        kw['ge'] = float.fromhex(rec['kw_hex']['ge'])
    return kw


def check_against_expect(got, exp, where=''):
    """`got`: dict in the oracle/ref_driver format.  `exp`: a golden record's expect block."""
    assert got['init_rc'] == exp['init_rc'], (where, 'init_rc', got['init_rc'], exp['init_rc'])
    if 'band' in exp:
        assert tuple(got['band']) == tuple(exp['band']), (where, 'band', got['band'], exp['band'])
    if exp['init_rc'] != 0:
        return
    assert tuple(got['opt']) == tuple(exp['opt']), (where, 'opt', got['opt'], exp['opt'])
    if 'num_rows' in got and got['num_rows'] is not None:
        assert got['num_rows'] == exp['num_rows'], (where, 'num_rows')
    if 'score' not in exp:
        return
    if 'score_hex' in exp:
        assert float(got['score']) == float.fromhex(exp['score_hex']), \
            (where, 'score', got['score'], exp['score'])
    else:
        assert got['score'] == exp['score'], (where, 'score', got['score'], exp['score'])
    if exp['panick']:
        if got.get('would_panick') is not None:
            assert got['would_panick'], (where, 'panick expected')
        return
    if got.get('would_panick') is not None:
        assert not got['would_panick'], (where, 'unexpected panick')
    assert bool(got['tb_null']) == bool(exp['tb_null']), (where, 'tb_null', got['tb_null'], exp['tb_null'])
    if exp['tb_null']:
        return
    tx = got['transcript']
    assert len(tx) == exp['tx_len'], (where, 'tx_len', len(tx), exp['tx_len'])
    assert hashlib.sha256(tx.encode()).hexdigest() == exp['tx_sha256'], (where, 'transcript', tx[:60])
    if 'transcript' in exp:
        assert tx == exp['transcript'], (where, tx, exp['transcript'])
    assert got['origin_idx'] == exp['origin_idx'], (where, 'origin_idx', got['origin_idx'], exp['origin_idx'])
    assert got['mutant_idx'] == exp['mutant_idx'], (where, 'mutant_idx', got['mutant_idx'], exp['mutant_idx'])
