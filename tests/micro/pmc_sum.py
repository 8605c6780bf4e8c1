"""Sums rocprofv3 --pmc counter CSVs per kernel: python tests/micro/pmc_sum.py <dir> [kernel substring]"""
import collections
import csv
import glob
import os
import sys

d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ''
tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = r.get('Kernel_Name', '')
            if sub and sub not in k:
                continue
            tot[k][r['Counter_Name']] += float(r['Counter_Value'])
            cnt[k].add(r.get('Dispatch_Id'))
for k in tot:
    print('kernel %s: %d dispatches' % (k[:90], len(cnt[k])))
    for c in sorted(tot[k]):
        print('  %-22s total %.6g  per dispatch %.6g' % (c, tot[k][c], tot[k][c] / max(1, len(cnt[k]))))
