#!/usr/bin/env python3
"""Compare two rocprofv3 kernel_stats.csv files per kernel (per step): tools/cmp_stats.py A.csv B.csv [steps]"""
import csv, re, sys
def load(f):
    d = {}
    for r in csv.DictReader(open(f)):
        n = re.sub(r'^_Z\d+', '', r['Name'])[:72]
        d[n] = (int(r['Calls']), float(r['TotalDurationNs']))
    return d
a, b = load(sys.argv[1]), load(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 32
print('total ms/step %.4f -> %.4f' % (sum(v[1] for v in a.values()) / steps / 1e6, sum(v[1] for v in b.values()) / steps / 1e6))
for k in sorted(set(a) | set(b), key=lambda k: -(a.get(k, (0, 0))[1] + b.get(k, (0, 0))[1])):
    ca, da = a.get(k, (0, 0)); cb, db = b.get(k, (0, 0))
    if abs(da - db) / steps / 1e3 > 1.0 or ca != cb:
        print(f"{k:72s} calls {ca/steps:5.1f}->{cb/steps:5.1f}  us/step {da/steps/1e3:8.1f} -> {db/steps/1e3:8.1f}  avg {da/max(ca,1)/1e3:6.1f}->{db/max(cb,1)/1e3:6.1f}")
