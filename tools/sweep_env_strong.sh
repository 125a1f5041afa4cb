#!/bin/bash
# One build, values of one environment variable, on the strong-split proxy (128-spp share and whole frame).
# usage: tools/sweep_env_strong.sh VAR v1 v2 ...
cd $GRAFT_REPO_ROOT
var=$1; shift
for rep in 1 2; do
for v in "$@"; do
  export $var=$v
  echo "== $var=$v"
  timeout -k 10 300 python3 tools/strong_proxy.py 2>&1 | grep "128 spp\|1024 spp\|at 8 GPUs" || exit 1
done
done
