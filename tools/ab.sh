#!/bin/bash
# A/B of builds of libmrt_hip.so on one box: raytracer_amd/ab/<name>.so for every name given, interleaved.
# usage: tools/ab.sh [-w workload] A B [C ...]
cd $GRAFT_REPO_ROOT
wl=million_bunny
if [ "$1" = "-w" ]; then wl=$2; shift 2; fi
for rep in 1 2; do
for v in "$@"; do
  cp raytracer_amd/ab/$v.so raytracer_amd/libmrt_hip.so
  timeout -k 10 300 python3 bench.py --workload $wl --no-secondary --steps 2 --warmup 1 --no-cpu $AB_FLAGS 2>&1 | tail -1 | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); print('$v', '$wl', round(r['value'],1), round(r['ms_per_step'],1), {k:round(x,1) for k,x in r['kernel_ms_per_step'].items()})" || exit 1
done
done
