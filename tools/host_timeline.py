#!/usr/bin/env python3
"""Host launch calls beside the kernels they start, for the last steps of a rocprofv3 --hip-trace --kernel-trace run:
   tools/host_timeline.py <dir with *_hip_api_trace.csv and *_kernel_trace.csv> [N last kernels]
Per kernel: gap to the previous kernel's end, and how long before its start the host's launch call had returned (negative lead =
the GPU waited for the host)."""
import csv
import glob
import re
import sys

d = sys.argv[1]
nlast = int(sys.argv[2]) if len(sys.argv) > 2 else 40
api = {}
for f in glob.glob(d + "/**/*hip_api_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        api[r["Correlation_Id"]] = (r["Function"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]))
ks = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("sphx::", "")
        ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k, r["Correlation_Id"]))
ks.sort()
ks = ks[-nlast:]
t0 = ks[0][0]
print(f"{'kernel':40s} {'start_us':>9s} {'dur_us':>7s} {'gap_us':>7s} {'launch_call_us':>14s} {'call_dur':>8s} {'lead_us':>8s}")
prev_end = None
for s, e, k, cid in ks:
    a = api.get(cid)
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    if a:
        print(f"{k[:40]:40s} {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} {gap:7.2f} {(a[1] - t0) / 1e3:14.1f} {(a[2] - a[1]) / 1e3:8.2f} {(s - a[2]) / 1e3:8.1f}")
    else:
        print(f"{k[:40]:40s} {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} {gap:7.2f} {'?':>14s}")
    prev_end = e
