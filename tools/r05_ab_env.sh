#!/bin/bash
# GPU box: A/B of environment switches of ONE library under the profiler (16 M kernel durations) + plain bench lines.
#   tools/r05_ab_env.sh OUTNAME "NAME=ENV=VALUE ..." ...      e.g.  tools/r05_ab_env.sh x "base=" "pair=SPHX_PAIR=1"
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline --no-also --prewarm-ms 0"
names=""
for spec in "$@"; do
  name=${spec%%=*}; envs=${spec#*=}; names="$names $name"
  for P in 16000000 1000000; do
    env $envs timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_${P}_$name -- $B --steps $([ $P = 1000000 ] && echo 100 || echo 20) --warmup 2 --particles $P > $out/stats_${P}_$name.log 2>&1; echo "$name $P stats rc=$?"
    f=$(find $out/stats_${P}_$name -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python3 $GRAFT_REPO_ROOT/tools/summarize_profile.py $f > $out/stats_${P}_$name.txt
    find $out -name "*.csv" -delete; find $out -type d -empty -delete
  done
done
cd $GRAFT_REPO_ROOT
for round in 1 2; do for spec in "$@"; do
  name=${spec%%=*}; envs=${spec#*=}
  for P in 16000000 1000000; do
    env $envs timeout 300 python3 bench.py --steps $([ $P = 1000000 ] && echo 100 || echo 20) --particles $P --no-cpu-baseline --no-also --no-roofline > $out/bench_${P}_${name}_$round.json 2> $out/bench_${P}_${name}_$round.err
    python3 - $out/bench_${P}_${name}_$round.json $name $P $round <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print(sys.argv[2], sys.argv[3], "round", sys.argv[4], round(d['value']/1e9,3), "G/s", round(d['ms_per_step'],4), "ms")
PY
  done
done; done
python3 - $out $names <<'PY'
import sys, re, os
out = sys.argv[1]; vs = sys.argv[2:]
for P in ("16000000", "1000000"):
    rows = {}
    for v in vs:
        fn = f"{out}/stats_{P}_{v}.txt"
        if not os.path.exists(fn): continue
        for l in open(fn).read().split("\n")[1:]:
            m = re.match(r"(.{48}) +(\d+) +([\d.]+) +([\d.]+) +([\d.]+)", l)
            if m: rows.setdefault(m.group(1).strip(), {})[v] = (int(m.group(2)), float(m.group(3)))
    print(f"\n== kernel durations at {P} particles (rocprofv3 --stats, avg us) ==")
    print("%-44s" % "kernel" + "".join("%12s" % v[:11] for v in vs))
    for k, r in sorted(rows.items(), key=lambda kv: -max(x[1] * x[0] for x in kv[1].values())):
        if max(x[0] for x in r.values()) < 15: continue
        print("%-44s" % k[:44] + "".join("%12.2f" % r[v][1] if v in r else "%12s" % "-" for v in vs))
PY
