#!/usr/bin/env python3
"""GPU idle time inside a train step from a rocprofv3 --kernel-trace CSV: union of kernel intervals vs span."""
import csv, sys, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows)
n = len(iv); seg = iv[int(n * float(sys.argv[2]) if len(sys.argv) > 2 else n // 2):]
t0, t1 = seg[0][0], max(e for s, e in seg)
busy = 0; cs, ce = seg[0]; gaps = []
for s, e in seg[1:]:
    if s > ce:
        busy += ce - cs; gaps.append(s - ce); cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
print(f"kernels {len(seg)}  span {(t1-t0)/1e6:.1f} ms  busy(union) {busy/1e6:.1f} ms  idle {100*(1-busy/(t1-t0)):.1f} %")
print(f"gaps: {len(gaps)}  median {statistics.median(gaps)/1e3:.1f} us  mean {sum(gaps)/len(gaps)/1e3:.1f} us  total {sum(gaps)/1e6:.1f} ms")
big = sorted(gaps, reverse=True)[:10]
print("largest gaps (us):", [round(g / 1e3) for g in big])
