#!/bin/bash
# Counter passes on the vector-memory pipeline (address unit, L1, L1 -> L2 requests and latency) over
# one step of one workload.   usage: tools/profile_mem_pipe.sh <outdir under gpurun_out> [workload]
out=${1:-gpurun_out/prof_mem}
wl=${2:-million_bunny}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
short="--workload $wl --no-secondary --serial --steps 1 --warmup 0 --no-cpu"
timeout -s KILL 400 rocprofv3 --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD --output-format csv -d $out/p1 -- python3 bench.py $short > $out/p1.log 2>&1 || { echo "pass 1 failed"; tail -5 $out/p1.log; exit 1; }
echo "pass 1 done"
timeout -s KILL 400 rocprofv3 --pmc TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS --output-format csv -d $out/p2 -- python3 bench.py $short > $out/p2.log 2>&1 || { echo "pass 2 failed"; tail -5 $out/p2.log; exit 1; }
echo all done
