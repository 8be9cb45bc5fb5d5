#!/bin/bash
# kernel-trace statistics of the late window (after 3750 steps) for the product library and variants
out=$GRAFT_REPO_ROOT/gpurun_out/late_stats; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for v in base "$@"; do
  if [ $v = base ]; then unset SPHX_LIB; else export SPHX_LIB=$GRAFT_REPO_ROOT/yasph2d_amd/variants/libsphx_$v.so; fi
  timeout 400 rocprofv3 --kernel-trace --output-format csv -d $out/kt_$v -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline --no-also --prewarm-ms 0 --steps 200 --warmup 0 --skip-steps 3750 > $out/kt_$v.log 2>&1; echo "$v rc=$?"
  f=$(find $out/kt_$v -name "*kernel_trace.csv" | head -1)
  python3 - "$f" > $out/kt_$v.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last 200 steps: take kernels after the last 200 occurrences of k_nonpressure
idx = [k for k, r in enumerate(rows) if 'k_nonpressure' in r['Kernel_Name']]
start = idx[-201] if len(idx) > 201 else 0
end = idx[-1]
sel = rows[start:end]
d = collections.defaultdict(lambda: [0, 0])
for r in sel:
    n = r['Kernel_Name'].split('(')[0][:60]
    d[n][0] += 1; d[n][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
busy = sum(v[1] for v in d.values()); span = int(sel[-1]['End_Timestamp']) - int(sel[0]['Start_Timestamp'])
print('steps 200 span_us_per_step %.1f busy_us_per_step %.1f' % (span / 200e3, busy / 200e3))
for n, (c, t) in sorted(d.items(), key=lambda x: -x[1][1]):
    print('%-62s n/step %.2f  us/launch %.2f  us/step %.2f' % (n, c / 200, t / c / 1e3, t / 200e3))
for key in ('k_compute_error<false', 'k_predict', 'k_nonpressure'):
    ds = sorted((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in sel if key in r['Kernel_Name'])
    if ds: print(key, 'min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f' % (ds[0], ds[len(ds)//10], ds[len(ds)//2], ds[len(ds)*9//10], ds[-1]))
for k, r in enumerate(sel[:40]):
    print('%-40s start+%.1f dur %.1f' % (r['Kernel_Name'].split('(')[0][:40], (int(r['Start_Timestamp']) - int(sel[0]['Start_Timestamp'])) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
PY
  find $out/kt_$v -name "*.csv" -delete
  echo "== $v"; cat $out/kt_$v.txt
done
