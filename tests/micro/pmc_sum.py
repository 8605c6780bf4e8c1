"""Sums rocprofv3 --pmc counter CSVs per kernel:

    python tests/micro/pmc_sum.py <dir> [kernel substring] [--json OUT --library-name NAME --pairs N]

With --json the per-launch means of the (single) matching kernel are written in the format bench.py reads
(profiles/pmc_kernel.json): counters, HBM bytes per launch (WRITE_SIZE + 2 x FETCH_SIZE, KiB -> bytes: gfx950 reports half
the bytes of a wide coalesced read, MI355X_MICROARCH.md), and the fingerprint of the kernel's code at collection time
(biseqt_amd/pwlib/kernel_hashes.json), which ties the numbers to one code object."""
import collections
import csv
import glob
import json
import os
import sys

args = [a for a in sys.argv[1:] if not a.startswith('--')]
opts = {}
it = iter(sys.argv[1:])
for a in it:
    if a.startswith('--'):
        opts[a[2:]] = next(it)
args = [a for a in args if a not in opts.values()]
d = args[0]
sub = args[1] if len(args) > 1 else ''
tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = r.get('Kernel_Name', '')
            if sub and sub not in k:
                continue
            tot[k][r['Counter_Name']] += float(r['Counter_Value'])
            cnt[k][r['Counter_Name']].add((f, r.get('Dispatch_Id')))
per = {}
for k in tot:
    n = max(len(v) for v in cnt[k].values())
    print('kernel %s: %d dispatches per pass' % (k[:90], n))
    per[k] = {}
    for c in sorted(tot[k]):
        per[k][c] = tot[k][c] / max(1, len(cnt[k][c]))
        print('  %-22s total %.6g  per dispatch %.6g' % (c, tot[k][c], per[k][c]))
if 'json' in opts:
    assert len(per) == 1, 'need exactly one matching kernel, got %r' % list(per)
    (k, c), = per.items()
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
    with open(os.path.join(root, 'biseqt_amd', 'pwlib', 'kernel_hashes.json')) as f:
        hashes = json.load(f)
    symbol = next((s for s in hashes if s.replace('void ', '').startswith(k.replace('void ', '').split('(')[0])), None)
    rec = {'kernel': opts.get('library-name', k), 'symbol': symbol, 'pairs': int(opts.get('pairs', 0)),
           'code_sha256': hashes.get(symbol), 'dispatches_averaged': max(len(v) for v in cnt[k].values())}
    rec.update({name: round(v, 1) for name, v in c.items()})
    if 'WRITE_SIZE' in c and 'FETCH_SIZE' in c:
        rec['hbm_bytes_per_launch'] = int((c['WRITE_SIZE'] + 2 * c['FETCH_SIZE']) * 1024)
    rec['note'] = ('rocprofv3 --kernel-trace --pmc <set>, one run per counter set, on `bench.py --inflight 1 --steps 2 --warmup 1 '
                   '--no-cpu-baseline --no-extras --min-seconds 0` (tests/micro/profile_bench.sh); per launch of the fill kernel; '
                   'WRITE_SIZE / FETCH_SIZE in KiB, FETCH_SIZE doubled in hbm_bytes_per_launch per MI355X_MICROARCH.md')
    with open(opts['json'], 'w') as f:
        json.dump(rec, f, indent=1)
    print('wrote', opts['json'], 'symbol', symbol)
