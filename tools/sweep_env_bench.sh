#!/bin/bash
# One build, one workload, a list of values of one environment variable: whole-frame times.
# usage: tools/sweep_env_bench.sh <so name under raytracer_amd/ab | -> <workload> VAR v1 v2 ...
cd $GRAFT_REPO_ROOT
so=$1; wl=$2; var=$3; shift 3
[ "$so" != "-" ] && cp raytracer_amd/ab/$so.so raytracer_amd/libmrt_hip.so
for v in "$@"; do
  export $var=$v
  timeout -k 10 300 python3 bench.py --workload $wl --no-secondary --no-strong-proxy --no-cpu --steps 3 --warmup 1 2>&1 | tail -1 | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); print('$so $var=$v', '$wl', round(r['value'],1), round(r['ms_per_step'],1), {k:round(x,1) for k,x in r['kernel_ms_per_step'].items()})" || exit 1
done
