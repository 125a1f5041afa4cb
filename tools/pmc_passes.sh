#!/bin/bash
# Hardware-counter passes over a short bench run (one pass per rocprofv3 invocation).
# usage: tools/pmc_passes.sh <outdir>
out=${1:-gpurun_out/pmc}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() {
  name=$1; shift
  timeout -s KILL 300 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -- python3 bench.py --steps 1 --warmup 0 --no-cpu --spp-y 1 > $out/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $out/$name.log; exit 1; }
  echo "pass $name done"
}
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU &&
run sq2 SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_INSTS_FLAT &&
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum &&
run fetch FETCH_SIZE &&
run write WRITE_SIZE &&
run tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
