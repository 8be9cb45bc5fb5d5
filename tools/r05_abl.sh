#!/bin/bash
# GPU box: kernel durations at 16 M (rocprofv3 --stats) of ablation variants (results wrong, timing informative).  tools/r05_abl.sh OUTNAME VARIANT...
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline --no-also --prewarm-ms 0"
# (one discarded run first: the first run of a call on a fresh box is 1-2 % slower than the following ones)
timeout 400 $B --steps 10 --warmup 2 ${ABL_ARGS} > /dev/null 2>&1
for v in "$@"; do
  if [ $v = base ]; then unset SPHX_LIB; else export SPHX_LIB=$GRAFT_REPO_ROOT/yasph2d_amd/variants/libsphx_$v.so; fi
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats16_$v -- $B --steps 10 --warmup 2 ${ABL_ARGS} > $out/stats16_$v.log 2>&1; echo "$v stats16 rc=$?"
  f=$(find $out/stats16_$v -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python3 $GRAFT_REPO_ROOT/tools/summarize_profile.py $f > $out/stats_16M_$v.txt
  find $out -name "*.csv" -delete; find $out -type d -empty -delete
done
python3 - $out "$@" <<'PY'
import sys, re, os
out = sys.argv[1]; vs = sys.argv[2:]
rows = {}
for v in vs:
    fn = f"{out}/stats_16M_{v}.txt"
    if not os.path.exists(fn): continue
    for l in open(fn).read().split("\n")[1:]:
        m = re.match(r"(.{48}) +(\d+) +([\d.]+) +([\d.]+) +([\d.]+)", l)
        if m: rows.setdefault(m.group(1).strip(), {})[v] = (int(m.group(2)), float(m.group(3)))
print("%-40s" % "kernel (avg us at 16 M)" + "".join("%12s" % v[:11] for v in vs))
for k, r in sorted(rows.items(), key=lambda kv: -max(x[1] * x[0] for x in kv[1].values())):
    if max(x[0] for x in r.values()) < 8: continue
    print("%-40s" % k[:40] + "".join("%12.2f" % r[v][1] if v in r else "%12s" % "-" for v in vs))
PY
