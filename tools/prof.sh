#!/bin/bash
# GPU box: every profiling pass of the rounds in one script (round 6 folded tools/r05_*.sh into it).
#   tools/prof.sh stats  OUT VARIANT...   kernel durations (rocprofv3 --kernel-trace --stats), one column per variant, same box
#   tools/prof.sh tail   OUT N VARIANT... the same over the LAST N dispatches only (the timed window of a --skip-steps run)
#   tools/prof.sh stamps OUT VARIANT...   in-kernel phase stamps of variants built with -DSPHX_STAMPS
#   tools/prof.sh bench  OUT VARIANT...   un-profiled bench.py runs, alternating (PROF_REPEAT rounds): us per step
#   tools/prof.sh pmc    OUT NAME COUNTER...  one counter pass (own run, --kernel-trace only, as MI355X_MICROARCH.md prescribes)
#   tools/prof.sh env    OUT "A=1 B=2" ...    stats columns for environment settings of the PRODUCT library (name=ENVSTRING)
# VARIANT = base (the product library) or NAME of yasph2d_amd/variants/libsphx_NAME.so (tools/ab_build.sh NAME -D...).
# PROF_ARGS = extra bench.py arguments (default workload: 16 M; e.g. "--particles 1000000", "--skip-steps 2500").
mode=$1; out=$GRAFT_REPO_ROOT/gpurun_out/$2; shift 2
mkdir -p $out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-roofline --no-also --prewarm-ms 0 --steps ${PROF_STEPS:-10} --warmup 2 ${PROF_ARGS}"
lib() { if [ "$1" = base ]; then unset SPHX_LIB; else export SPHX_LIB=$R/yasph2d_amd/variants/libsphx_$1.so; fi; }
table() {  # columns of per-kernel averages from $out/NAME.txt files
python3 - $out "$@" <<'PY'
import sys, re, os
out = sys.argv[1]; vs = sys.argv[2:]
rows = {}
for v in vs:
    fn = f"{out}/{v}.txt"
    if not os.path.exists(fn): continue
    for l in open(fn).read().split("\n")[1:]:
        m = re.match(r"(.{40,48}?) +(\d+) +([\d.]+)", l)
        if m: rows.setdefault(m.group(1).strip(), {})[v] = (int(m.group(2)), float(m.group(3)))
print("%-40s" % "kernel (avg us)" + "".join("%12s" % v[:11] for v in vs))
for k, r in sorted(rows.items(), key=lambda kv: -max(x[1] * x[0] for x in kv[1].values())):
    if max(x[0] for x in r.values()) < 3: continue
    print("%-40s" % k[:40] + "".join("%12.2f" % r[v][1] if v in r else "%12s" % "-" for v in vs))
PY
}
case $mode in
stats|env)
  timeout 400 $B > /dev/null 2>&1   # (one discarded run: the first run of a call on a fresh box is 1-2 % slower)
  names=()
  for v in "$@"; do
    if [ $mode = env ]; then n=${v%%=*}; e=${v#*=}; names+=($n); else n=$v; e=""; lib $v; names+=($n); fi
    env $e timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/d_$n -- $B > $out/$n.log 2>&1; echo "$n rc=$?"
    f=$(find $out/d_$n -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python3 $R/tools/summarize_profile.py $f > $out/$n.txt
    rm -rf $out/d_$n
  done
  table "${names[@]}" ;;
tail)
  N=$1; shift
  timeout 400 $B > /dev/null 2>&1
  names=()
  for vv in "$@"; do
    if [[ "$vv" == *=* ]]; then v=${vv%%=*}; e=${vv#*=}; lib base; else v=$vv; e=""; lib $vv; fi   # NAME=ENV... : the product library under that environment
    names+=($v)
    env $e timeout 900 rocprofv3 --kernel-trace --output-format csv -d $out/d_$v -- $B > $out/$v.log 2>&1; echo "$v rc=$?"
    f=$(find $out/d_$v -name "*kernel_trace.csv" | head -1)
    [ -n "$f" ] && python3 - $f $N > $out/$v.txt <<'PY'
import csv, re, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("sphx::", "")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k))
rows.sort(); rows = rows[-int(sys.argv[2]):]
d, c = collections.Counter(), collections.Counter()
for s, e, k in rows: d[k] += e - s; c[k] += 1
print(f"{'kernel':48s} {'calls':>7s} {'avg_us':>9s}   (last {len(rows)} dispatches, span {(rows[-1][1] - rows[0][0]) / 1e3:.1f} us)")
for k in sorted(d, key=lambda k: -d[k]): print(f"{k[:48]:48s} {c[k]:7d} {d[k] / c[k] / 1e3:9.2f}")
PY
    rm -rf $out/d_$v
  done
  table "${names[@]}" ;;
stamps)
  cd $R
  for v in "$@"; do
    lib $v
    timeout 600 $B --steps 20 > $out/bench_$v.json 2> $out/bench_$v.err
    echo "== $v"; grep SPHX_STAMPS $out/bench_$v.err | tail -4
  done ;;
bench)   # un-profiled step time, variants alternating, PROF_REPEAT rounds (default 3): what the driver's clock sees
  for r in $(seq 1 ${PROF_REPEAT:-3}); do
    for vv in "$@"; do
      if [[ "$vv" == *=* ]]; then v=${vv%%=*}; e=${vv#*=}; lib base; else v=$vv; e=""; lib $vv; fi
      env $e $B --steps ${PROF_STEPS:-100} --warmup 10 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step']*1000,1), 'us/step', round(d['value']/1e9,3), 'G/s')" | tee -a $out/bench.txt
    done
  done ;;
pmc)
  name=$1; shift
  timeout 900 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/d_$name -- $B > $out/$name.log 2>&1; echo "$name rc=$?"
  f=$(find $out/d_$name -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python3 $R/tools/pmc_table.py $f > $out/$name.txt
  rm -rf $out/d_$name; head -14 $out/$name.txt ;;
esac
