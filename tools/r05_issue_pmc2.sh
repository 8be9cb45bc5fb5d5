#!/bin/bash
# which SQ counter tracks the TIME a SIMD spends on vector instructions (not the issue slots)?  tools/r05_issue_pmc2.sh OUTNAME
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out; cd $GRAFT_REPO_ROOT
cp tools/bin/valu_issue_bench /tmp/valu_issue_bench
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $out/pmc -- /tmp/valu_issue_bench > $out/pmc.log 2>&1; echo "pmc rc=$?"
f=$(find $out/pmc -name "*counter_collection.csv" | head -1)
python3 - $f > $out/issue_rate_pmc2.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.OrderedDict()
for r in rows:
    k = (r["Dispatch_Id"], r["Kernel_Name"], r["Grid_Size"])
    by.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"])
print("%-30s %8s %12s %10s %14s %16s %14s %12s" % ("kernel", "waves", "INSTS/wave", "ACT/INSTS", "THREAD_CYC/INST", "BUSY/32/inst/SIMD", "ACT_ANY/INSTS", "WAVE_CYC/inst"))
for (d, name, grid), c in by.items():
    w = c.get("SQ_WAVES", 0)
    if not w: continue
    iv = c.get("SQ_INSTS_VALU", 0)
    if iv / w < 20000: continue
    if int(grid) != 524288: continue
    per_simd = iv / 1024.0
    print("%-30s %8d %12.0f %10.3f %14.3f %16.3f %14.3f %12.3f" % (name[:30], w, iv / w, c.get("SQ_ACTIVE_INST_VALU", 0) / iv, c.get("SQ_THREAD_CYCLES_VALU", 0) / iv,
          c.get("SQ_BUSY_CYCLES", 0) / 32.0 / per_simd, c.get("SQ_ACTIVE_INST_ANY", 0) / iv, c.get("SQ_WAVE_CYCLES", 0) / iv))
PY
find $out/pmc -name "*.csv" -delete; find $out -type d -empty -delete
cat $out/issue_rate_pmc2.txt
