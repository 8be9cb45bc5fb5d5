#!/bin/bash
# GPU box: same-box A/B of an environment switch of libsphx.  tools/r04_env_ab.sh OUTNAME VAR  (VAR=0 against VAR=1, alternating, 3 repetitions)
out=$GRAFT_REPO_ROOT/gpurun_out/$1; var=$2; mkdir -p $out; cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for val in 0 1; do for P in 16000000 1000000; do
  env $var=$val timeout 300 python bench.py --steps $([ $P = 1000000 ] && echo 200 || echo 40) --particles $P --no-cpu-baseline --no-also > $out/b_${val}_${P}_$rep.json 2>/dev/null
  python3 - $out/b_${val}_${P}_$rep.json "$var=$val" $P <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print(sys.argv[2], sys.argv[3].ljust(9), round(d['value']/1e9,3), round(d['ms_per_step'],4), {k[:14]:round(v*1000,1) for k,v in d['roofline']['per_kernel_ms_per_step_event_inflated'].items()})
PY
done; done; done
