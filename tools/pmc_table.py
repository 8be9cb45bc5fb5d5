#!/usr/bin/env python3
"""Print per-kernel means of the counters in a rocprofv3 counter_collection.csv (one column per counter)."""
import collections
import csv
import re
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.Counter())
names = []
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("sphx::", "")
    c = r["Counter_Name"]
    if c not in names:
        names.append(c)
    acc[k][c] += float(r["Counter_Value"])
    cnt[k][c] += 1
filt = sys.argv[2] if len(sys.argv) > 2 else ""
print(f"{'kernel':36s} " + " ".join(f"{n[3:][:13]:>13s}" for n in names))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][names[0]]):
    if filt in k:
        print(f"{k[:36]:36s} " + " ".join(f"{v[x] / max(cnt[k][x], 1):13.0f}" for x in names))
