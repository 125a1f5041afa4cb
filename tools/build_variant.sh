#!/bin/bash
# Builds libmrt_hip.so variants for tools/ab.sh: raytracer_amd/ab/<name>.so from the current tree
# (extra compiler flags after the name), or from the tree with some files taken from a git revision:
#   tools/build_variant.sh new
#   tools/build_variant.sh base --from HEAD raytracer_amd/csrc/device_render.hpp raytracer_amd/csrc/render_host.inc
#   tools/build_variant.sh lds8 -DMRT_STACK_LDS=8
set -e
cd "$(dirname "$0")/.."
name=$1; shift
root=.
if [ "$1" = "--from" ]; then
  rev=$2; shift 2
  root=$(mktemp -d)
  mkdir -p $root/raytracer_amd
  cp -r raytracer_amd/csrc $root/raytracer_amd/csrc
  cp -r include $root/include
  for f in "$@"; do git show $rev:$f > $root/$f; done
  set --
fi
mkdir -p raytracer_amd/ab
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -std=c++17 -fPIC -shared -I$root/include "$@" -o raytracer_amd/ab/$name.so $root/raytracer_amd/csrc/mrt_hip.hip
echo built raytracer_amd/ab/$name.so
