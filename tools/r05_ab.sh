#!/bin/bash
# GPU box: same-box A/B of library variants under the profiler.  tools/r05_ab.sh OUTNAME [VARIANT...]
#   per variant ("base" = the product library, others = yasph2d_amd/variants/libsphx_NAME.so):
#   rocprofv3 --kernel-trace --stats at 16 M (kernel durations), SQ counters at 1 M (instructions per wavefront), plain bench at 16 M / 1 M
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline --no-also --prewarm-ms 0"
SQ="--pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"
for v in base "$@"; do
  if [ $v = base ]; then unset SPHX_LIB; else export SPHX_LIB=$GRAFT_REPO_ROOT/yasph2d_amd/variants/libsphx_$v.so; fi
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats16_$v -- $B --steps 20 --warmup 2 > $out/stats16_$v.log 2>&1; echo "$v stats16 rc=$?"
  f=$(find $out/stats16_$v -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python3 $GRAFT_REPO_ROOT/tools/summarize_profile.py $f > $out/stats_16M_$v.txt
  timeout 400 rocprofv3 $SQ --kernel-trace --output-format csv -d $out/sq1_$v -- $B --steps 20 --warmup 2 --particles 1000000 > $out/sq1_$v.log 2>&1; echo "$v sq1 rc=$?"
  f=$(find $out/sq1_$v -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python3 $GRAFT_REPO_ROOT/tools/pmc_table.py $f > $out/sq_1M_$v.txt
  find $out -name "*.csv" -delete; find $out -type d -empty -delete
done
cd $GRAFT_REPO_ROOT
# plain bench lines, alternating (clock drift between variants shows up as a difference between the two rounds)
for round in 1 2; do for v in base "$@"; do
  if [ $v = base ]; then unset SPHX_LIB; else export SPHX_LIB=$GRAFT_REPO_ROOT/yasph2d_amd/variants/libsphx_$v.so; fi
  for P in 16000000 1000000; do
    timeout 300 python3 bench.py --steps $([ $P = 1000000 ] && echo 100 || echo 20) --particles $P --no-cpu-baseline --no-also --no-roofline > $out/bench_${P}_${v}_$round.json 2> $out/bench_${P}_${v}_$round.err
    python3 - $out/bench_${P}_${v}_$round.json $v $P $round <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print(sys.argv[2], sys.argv[3], "round", sys.argv[4], round(d['value']/1e9,3), "G/s", round(d['ms_per_step'],4), "ms")
PY
  done
done; done
python3 - $out base "$@" <<'PY'
import sys, re, os
out = sys.argv[1]; vs = sys.argv[2:]
def table(fn, col):
    d = {}
    if not os.path.exists(fn): return d
    for l in open(fn):
        p = l.split()
        if len(p) > col and p[0].startswith("k_"):
            try: d[" ".join(p[:len(p) - (len(p) - 1 if False else 0)]).split("  ")[0]] = p
            except Exception: pass
    return d
print("\n== kernel durations at 16 M (rocprofv3 --stats, avg us) ==")
rows = {}
for v in vs:
    fn = f"{out}/stats_16M_{v}.txt"
    if not os.path.exists(fn): continue
    for l in open(fn).read().split("\n")[1:]:
        m = re.match(r"(.{48}) +(\d+) +([\d.]+) +([\d.]+) +([\d.]+)", l)
        if m: rows.setdefault(m.group(1).strip(), {})[v] = (int(m.group(2)), float(m.group(3)))
print("%-48s" % "kernel" + "".join("%14s" % v for v in vs))
tot = {v: 0.0 for v in vs}
for k, r in sorted(rows.items(), key=lambda kv: -max(x[1] * x[0] for x in kv[1].values())):
    if max(x[0] for x in r.values()) < 15: continue
    print("%-48s" % k + "".join("%14.2f" % r[v][1] if v in r else "%14s" % "-" for v in vs))
    for v in vs:
        if v in r: tot[v] += r[v][1] * r[v][0] / 20.0
print("%-48s" % "sum per step (calls/20 x avg)" + "".join("%14.1f" % tot[v] for v in vs))
print("\n== vector instructions per wavefront at 1 M (SQ_INSTS_VALU / SQ_WAVES) ==")
rows = {}
for v in vs:
    fn = f"{out}/sq_1M_{v}.txt"
    if not os.path.exists(fn): continue
    lines = open(fn).read().split("\n")
    hdr = lines[0].split()
    for l in lines[1:]:
        m = re.match(r"(.{36})(.*)", l)
        if not m or not m.group(1).strip(): continue
        vals = m.group(2).split()
        if len(vals) != len(hdr) - 1: continue
        c = dict(zip(hdr[1:], map(float, vals)))
        if c.get("WAVES", 0) > 0: rows.setdefault(m.group(1).strip(), {})[v] = (c["INSTS_VALU"] / c["WAVES"], c.get("BUSY_CYCLES", 0))
print("%-40s" % "kernel" + "".join("%14s" % v for v in vs))
for k, r in rows.items():
    if not k.startswith("k_") : continue
    print("%-40s" % k + "".join("%14.1f" % r[v][0] if v in r else "%14s" % "-" for v in vs))
PY
