#!/bin/bash
# A/B of builds of libmrt_hip.so on the strong-split proxy (tools/strong_proxy.py): raytracer_amd/ab/<name>.so
# for every name given, interleaved.  usage: tools/ab_strong.sh [--only SPP] A B [C ...]
cd $GRAFT_REPO_ROOT
extra=""
if [ "$1" = "--only" ]; then extra="--only $2"; shift 2; fi
for rep in 1 2; do
for v in "$@"; do
  cp raytracer_amd/ab/$v.so raytracer_amd/libmrt_hip.so
  echo "== $v"
  timeout -k 10 300 python3 tools/strong_proxy.py $extra 2>&1 | grep -v amdgpu.ids || exit 1
done
done
