#!/bin/bash
# Hardware-counter passes over the traversal micro-benchmark (one pass per rocprofv3 invocation).
# usage: tools/pmc_trace.sh <outdir>
out=${1:-gpurun_out/pmc_trace}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -- python3 tools/trace_bench.py --reps 1 > $out/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $out/$name.log; exit 1; }
  echo "pass $name done"
}
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU &&
run sq2 SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM &&
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum &&
run tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
