#!/bin/bash
# A/B of one build with and without an environment variable on whole frames of a workload, interleaved.
# usage: tools/ab_env_bench.sh VAR=value [workload] [bench.py arguments]
cd $GRAFT_REPO_ROOT
var=$1; wl=${2:-million_bunny}; shift 2
for rep in 1 2; do
for v in off on; do
  if [ $v = on ]; then export "$var"; else unset "${var%%=*}"; fi
  echo "== $var $v"
  timeout -k 10 300 python3 bench.py --workload $wl --no-secondary --no-strong-proxy --no-cpu --steps 5 --warmup 2 "$@" 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j.get('roofline') or {}
        print(j['value'], 'Mrays/s', j['ms_per_step'], 'ms', j.get('kernel_ms_per_step'))
" || exit 1
done
done
