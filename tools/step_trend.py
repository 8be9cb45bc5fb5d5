#!/usr/bin/env python3
"""Per-step sums of kernel durations from a rocprofv3 kernel_trace.csv: how the cost of a step moves with the step index.
usage: step_trend.py kernel_trace.csv [anchor kernel substring, default k_nonpressure]"""
import csv, re, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("sphx::", "")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k))
rows.sort()
anchor = sys.argv[2] if len(sys.argv) > 2 else "k_nonpressure"
steps, cur = [], None
for s, e, k in rows:
    if anchor in k:
        cur = collections.Counter(); steps.append(cur)
    if cur is not None: cur[k] += (e - s) / 1e3
names = [k for k, _ in collections.Counter({k: v for st in steps for k, v in st.items()}).most_common(8)]
print("step " + " ".join(f"{n[:22]:>22s}" for n in names) + "      total")
for i, st in enumerate(steps):
    if i % 5 == 0 or i < 12: print(f"{i:4d} " + " ".join(f"{st.get(n, 0):22.1f}" for n in names) + f" {sum(st.values()):10.1f}")
