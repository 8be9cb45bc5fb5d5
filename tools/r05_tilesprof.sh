#!/bin/bash
# GPU box: kernel timeline of one step of the tile path at 16 M, world 1 (bench.py --force-tiles) — start / end / queue of every launch
out=$GRAFT_REPO_ROOT/gpurun_out/r05_tilesprof; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline --no-also --prewarm-ms 0"
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $out/tiles -- $B --steps 6 --warmup 2 --force-tiles > $out/tiles.log 2>&1; echo "tiles rc=$?"
f=$(find $out/tiles -name "*kernel_trace.csv" | head -1)
python3 - $f > $out/timeline_tiles.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
# the last 40 launches
for r in rows[-40:]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%-44s queue %s  start %10.1f us  end %10.1f us  dur %8.1f us" % (r["Kernel_Name"][:44], r.get("Queue_Id", "?"), s / 1e3, e / 1e3, (e - s) / 1e3))
PY
find $out -name "*.csv" -delete; find $out -type d -empty -delete
cat $out/timeline_tiles.txt
