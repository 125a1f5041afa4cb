#!/bin/bash
# Counter passes over single configurations of raytracer_amd/gather_bench (known access pattern: every
# lane reads one random 32-byte record, or 4 / 8 lanes one random 64- / 128-byte record), one counter
# set per rocprofv3 run, the program directly after "--".
# usage (on the GPU box): tools/gather_calibration.sh <outdir>
out=${1:-$GRAFT_REPO_ROOT/gpurun_out/gather_pmc}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
bin=$GRAFT_REPO_ROOT/raytracer_amd/gather_bench
pass() { # name, config words..., then "--", counters...
  name=$1; shift
  local c=()
  while [ "$1" != "--" ]; do c+=("$1"); shift; done
  shift
  tag=$(echo "${c[@]}" | tr ' ' '_')
  timeout -s KILL 90 rocprofv3 --pmc "$@" --output-format csv -d $out/$tag/$name -- $bin "${c[@]}" > $out/$tag.$name.log 2>&1 || { echo "pass $tag/$name failed"; tail -3 $out/$tag.$name.log; return 1; }
  grep -h "independent\|chained" $out/$tag.$name.log | tail -1
}
for cfg in "independent 2 6 32 2000" "independent 4096 6 32 2000" "independent 4096 6 128 2000" "independent 8 6 32 2000" "independent 8 6 128 2000" "independent 512 6 32 2000" "chained 4096 6 32 2000"; do
  pass fetch $cfg -- FETCH_SIZE &&
  pass tcc $cfg -- TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum &&
  pass ea $cfg -- TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum &&
  pass dram $cfg -- TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_32B_sum TCC_BUBBLE_sum &&
  pass tcp $cfg -- TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum || exit 1
done
