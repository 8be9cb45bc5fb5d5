#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/prof_*) into the small per-round summaries committed under profiles/.

usage: summarize_profile.py <kernel_stats.csv | counter_collection.csv> [--last N]   (--last: counter files, the last N dispatches only)
  kernel_stats        -> per-kernel calls / avg us / % table
  counter_collection  -> per-kernel mean counter value; FETCH_SIZE / WRITE_SIZE are reported in KiB by rocprofv3 and
                         FETCH_SIZE is doubled as MI355X_MICROARCH.md (HBM section) prescribes for gfx950.
"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"(sphx::)?([A-Za-z0-9_]+(<[^>]*>)?)", name)
    return m.group(2) if m else name[:40]


def main(path, last=0):
    rows = list(csv.DictReader(open(path)))
    if last and "Dispatch_Id" in rows[0]:  # the last `last` dispatches only (the timed window of a --skip-steps run)
        ids = sorted({int(r["Dispatch_Id"]) for r in rows})[-last:]
        keep = set(ids)
        rows = [r for r in rows if int(r["Dispatch_Id"]) in keep]
    if "TotalDurationNs" in rows[0]:
        print(f"{'kernel':48s} {'calls':>7s} {'avg_us':>9s} {'total_ms':>9s} {'pct':>6s}")
        for r in rows:
            print(f"{short(r['Name']):48s} {int(r['Calls']):7d} {float(r['AverageNs']) / 1e3:9.2f} {float(r['TotalDurationNs']) / 1e6:9.3f} {float(r['Percentage']):6.2f}")
    else:
        acc = defaultdict(lambda: [0, 0.0, 0.0])
        cname = rows[0]["Counter_Name"]
        for r in rows:
            a = acc[short(r["Kernel_Name"])]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
            a[2] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        corr = 2.0 if cname == "FETCH_SIZE" else 1.0
        print(f"counter {cname} (KiB per dispatch as reported; bytes column = KiB*1024*{corr:g} gfx950 correction)")
        print(f"{'kernel':48s} {'calls':>7s} {'mean_KiB':>12s} {'bytes/launch':>14s} {'avg_us':>9s}")
        for k, (n, v, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
            print(f"{k:48s} {n:7d} {v / n:12.1f} {v / n * 1024 * corr:14.0f} {t / n / 1e3:9.2f}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[sys.argv.index("--last") + 1]) if "--last" in sys.argv else 0)
