#!/bin/bash
# A/B of one build with and without an environment variable on the strong-split proxy, interleaved.
# usage: tools/ab_env_strong.sh [--only SPP] VAR=value
cd $GRAFT_REPO_ROOT
extra=""
if [ "$1" = "--only" ]; then extra="--only $2"; shift 2; fi
for rep in 1 2; do
for v in off on; do
  if [ $v = on ]; then export "$1"; else unset "${1%%=*}"; fi
  echo "== $1 $v"
  timeout -k 10 300 python3 tools/strong_proxy.py $extra 2>&1 | grep -v amdgpu.ids || exit 1
done
done
