#!/bin/bash
# GPU box: the whole -m gpu suite, then the default bench line.  tools/r04_full.sh OUTNAME
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $out
cd $GRAFT_REPO_ROOT
t0=$(date +%s)
timeout 2400 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; tail -3 $out/pytest.log
t0=$(date +%s)
timeout 600 python bench.py --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$? ($(( $(date +%s) - t0 )) s)"
python3 - $out/bench_default.json <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); r=d.get('roofline',{})
        print('HEAD', d['config']['workload'][:90]); print(round(d['value']/1e9,3),'G/s', round(d['ms_per_step'],4),'ms', 'bound',r.get('bound'),'frac',round(r.get('frac',0),4),'valu_issue_frac',r.get('valu_issue_frac'), 'avg_launch_ms', r.get('avg_launch_ms'))
        print({k[:22]:round(v*1000,1) for k,v in r.get('per_kernel_ms_per_step_event_inflated',{}).items()})
        for a in d.get('also',[]): print('ALSO', a['window'][:60], round(a['value']/1e9,3), round(a['ms_per_step'],4), a.get('dominant_kernel',{}).get('avg_launch_ms'))
        print('CPU', d.get('cpu_baseline',{}).get('value'))
PY
