#!/bin/bash
# Same-box A/B of ENVIRONMENT settings of the product library:  tools/ab_env.sh [-r ROUNDS] NAME=ENV ... (ENV may be empty: "base=")
cd $GRAFT_REPO_ROOT
rounds=3
if [ "$1" = "-r" ]; then rounds=$2; shift 2; fi
run() {  # name env label args
  env $2 python3 bench.py --no-cpu-baseline --no-roofline --no-also $4 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-10s %-9s' % ('$1', '$3'), round(d['ms_per_step']*1000,1), 'us/step', round(d['value']/1e9,3), 'G/s')"
}
run warm "" discard "--steps 30 --warmup 5" >/dev/null
for r in $(seq $rounds); do for v in "$@"; do run "${v%%=*}" "${v#*=}" "16M" "--steps 100 --warmup 10"; done; done
for r in $(seq $rounds); do for v in "$@"; do run "${v%%=*}" "${v#*=}" "1M" "--steps 300 --warmup 30 --particles 1000000"; done; done
for r in 1 2; do for v in "$@"; do run "${v%%=*}" "${v#*=}" "16M-late" "--steps 20 --warmup 2 --skip-steps 2500"; done; done
for r in 1 2; do for v in "$@"; do run "${v%%=*}" "${v#*=}" "1M-late" "--steps 100 --warmup 5 --particles 1000000 --skip-steps 3750"; done; done
