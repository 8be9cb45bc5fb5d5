#!/bin/bash
# kernel trace of the forced-tile path next to the single context: kernels per step, busy time and gaps
out=$GRAFT_REPO_ROOT/gpurun_out/tiletrace; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for v in single tiles; do
  a=""; [ $v = tiles ] && a="--force-tiles"
  timeout 400 rocprofv3 --kernel-trace --output-format csv -d $out/kt_$v -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline --no-also --prewarm-ms 0 --steps 200 --warmup 5 $a > $out/kt_$v.log 2>&1; echo "$v rc=$?"
  f=$(find $out/kt_$v -name "*kernel_trace.csv" | head -1)
  python3 - "$f" > $out/trace_$v.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [k for k, r in enumerate(rows) if 'k_nonpressure' in r['Kernel_Name']]
sel = rows[idx[-151]:idx[-1]]
d = collections.defaultdict(lambda: [0, 0])
gap = collections.defaultdict(lambda: [0, 0])
prev_end = None
for r in sel:
    n = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('sphx::', '')[:50]
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    d[n][0] += 1; d[n][1] += e - s
    if prev_end is not None: gap[n][0] += 1; gap[n][1] += max(0, s - prev_end)
    prev_end = max(prev_end or 0, e)
steps = 150
busy = sum(v[1] for v in d.values()); span = int(sel[-1]['End_Timestamp']) - int(sel[0]['Start_Timestamp'])
print('span_us_per_step %.1f busy %.1f gaps %.1f' % (span / steps / 1e3, busy / steps / 1e3, (span - busy) / steps / 1e3))
for n, (c, t) in sorted(d.items(), key=lambda x: -x[1][1]):
    print('%-52s n/step %.2f us/launch %6.2f us/step %6.2f  gap-before us/step %5.2f' % (n, c / steps, t / c / 1e3, t / steps / 1e3, gap[n][1] / steps / 1e3))
print('-- one step')
for r in rows[idx[-3]:idx[-2]]:
    print('%-46s +%7.1f dur %5.1f' % (r['Kernel_Name'].split('(')[0].replace('void ', '').replace('sphx::', '')[:46], (int(r['Start_Timestamp']) - int(rows[idx[-3]]['Start_Timestamp'])) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
PY
  find $out/kt_$v -name "*.csv" -delete
  echo "== $v"; cat $out/trace_$v.txt
done
