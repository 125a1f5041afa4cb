#!/bin/bash
# A/B of one build with and without an environment variable set, interleaved.
# usage: tools/ab_env.sh [-w workload] VAR=value
cd $GRAFT_REPO_ROOT
wl=million_bunny
if [ "$1" = "-w" ]; then wl=$2; shift 2; fi
for rep in 1 2; do
for v in off on; do
  if [ $v = on ]; then export "$1"; else unset "${1%%=*}"; fi
  timeout -k 10 300 python3 bench.py --workload $wl --no-secondary --steps 2 --warmup 1 --no-cpu 2>&1 | tail -1 | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); print('$v', '$wl', round(r['value'],1), round(r['ms_per_step'],1), {k:round(x,1) for k,x in r['kernel_ms_per_step'].items()})" || exit 1
done
done
