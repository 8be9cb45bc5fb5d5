#!/bin/bash
# GPU box: the issue-rate microbenchmark (tools/valu_issue_bench.hip), plain and under the SQ counters.  tools/r05_issue.sh OUTNAME
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $out
cd $GRAFT_REPO_ROOT
[ -x tools/bin/valu_issue_bench ] && cp tools/bin/valu_issue_bench /tmp/valu_issue_bench || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_issue_bench tools/valu_issue_bench.hip || exit 1
timeout 300 /tmp/valu_issue_bench > $out/issue_rate.txt 2>&1; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/pmc -- /tmp/valu_issue_bench > $out/pmc.log 2>&1; echo "pmc rc=$?"
f=$(find $out/pmc -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python3 - $f > $out/issue_rate_pmc.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
# one dispatch = one (kind, occupancy); the warm-up dispatch of a pair has 16 iterations, the measured one 256
by = collections.OrderedDict()
for r in rows:
    k = (r["Dispatch_Id"], r["Kernel_Name"], r["Grid_Size"], r.get("LDS_Block_Size", ""))
    by.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"])
print("%-60s %10s %8s %14s %14s %8s" % ("kernel", "grid", "waves", "INSTS_VALU/wave", "ACTIVE/INSTS", "busy"))
for (d, name, grid, lds), c in by.items():
    w = c.get("SQ_WAVES", 0)
    if not w: continue
    iv = c.get("SQ_INSTS_VALU", 0) / w
    if iv < 3000: continue   # warm-up dispatches
    print("%-60s %10s %8d %14.0f %14.3f %8.0f" % (name[:60], grid, w, iv, c.get("SQ_ACTIVE_INST_VALU", 0) / max(c.get("SQ_INSTS_VALU", 1), 1), c.get("SQ_BUSY_CYCLES", 0)))
PY
find $out/pmc -name "*.csv" -delete; find $out -type d -empty -delete
head -100 $out/issue_rate.txt
