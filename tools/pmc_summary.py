#!/usr/bin/env python3
import csv, sys, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        k = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])[:60]
        a = agg[k][r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
for k, v in agg.items():
    if 'wino' in k or 'igemm' in k or 'wgrad' in k:
        print(k)
        for c, x in sorted(v.items()):
            print(f"    {c:32s} {x[0]/x[1]:16.1f}  (n={x[1]})")
