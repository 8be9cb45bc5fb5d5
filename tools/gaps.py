#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace csv (one stream): per-kernel mean duration and the mean
gap that FOLLOWS each kernel, over the steady part of the trace."""
import collections
import csv
import re
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("sphx::", "")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k))
rows.sort()
skip = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 3
rows = rows[skip:]
dur, gap, cnt = collections.Counter(), collections.Counter(), collections.Counter()
for a, b in zip(rows, rows[1:]):
    dur[a[2]] += a[1] - a[0]
    gap[a[2]] += max(0, b[0] - a[1])
    cnt[a[2]] += 1
span = rows[-1][1] - rows[0][0]
busy = sum(e - s for s, e, _ in rows)
print(f"kernels {len(rows)}  span {span/1e3:.1f} us  busy {busy/1e3:.1f} us  idle {100*(1-busy/span):.1f} %")
print(f"{'kernel':40s} {'calls':>6s} {'avg_us':>8s} {'gap_after_us':>12s}")
for k in sorted(dur, key=lambda k: -dur[k]):
    print(f"{k[:40]:40s} {cnt[k]:6d} {dur[k]/cnt[k]/1e3:8.2f} {gap[k]/cnt[k]/1e3:12.2f}")
