#!/bin/bash
# duration of chosen kernels against the step index (kernel trace over a long run)
out=$GRAFT_REPO_ROOT/gpurun_out/drift; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for v in base "$@"; do
  if [ $v = base ]; then unset SPHX_LIB; else export SPHX_LIB=$GRAFT_REPO_ROOT/yasph2d_amd/variants/libsphx_$v.so; fi
  timeout 400 rocprofv3 --kernel-trace --output-format csv -d $out/kt_$v -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline --no-also --prewarm-ms 0 --steps 200 --warmup 0 --skip-steps 3750 > $out/kt_$v.log 2>&1; echo "$v rc=$?"
  f=$(find $out/kt_$v -name "*kernel_trace.csv" | head -1)
  python3 - "$f" > $out/drift_$v.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
step = -1
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in rows:
    n = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('sphx::', '')
    if n == 'k_nonpressure': step += 1
    if step < 0: continue
    a = acc[step // 250][n]; a[0] += 1; a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
names = ['k_nonpressure', 'k_predict', 'k_compute_error<false, true>', 'k_compute_error<false, false>', 'k_correct<false, true>', 'k_compute_error<true, false>', 'k_correct<false, false>', 'k_neighbor_build<3>', 'k_neighbor_build<2>', 'k_scan_onepass', 'k_rank_gather']
print('bucket ' + ' '.join('%s' % n[2:22] for n in names))
for b in sorted(acc):
    print('%5d ' % (b * 250) + ' '.join('%6.1f/%4.2f' % ((acc[b][n][1] / acc[b][n][0]) if acc[b][n][0] else 0, acc[b][n][0] / 250) for n in names))
PY
  find $out/kt_$v -name "*.csv" -delete
  echo "== $v"; cat $out/drift_$v.txt
done
