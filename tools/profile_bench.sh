#!/bin/bash
# Profiles of the benchmark command for profiles/: kernel trace + stats of the default command, then
# one hardware-counter pass per rocprofv3 run (FETCH_SIZE and WRITE_SIZE do not fit one pass) over
# one step of one workload.
# usage: tools/profile_bench.sh <outdir under gpurun_out> [workload] [camera rays as packets: 0 | 1]
out=${1:-gpurun_out/prof_bench}
wl=${2:-million_bunny}
# how the camera rays are traced is fixed (0 = streaming traversal, 1 = packets), so that the launches with
# which mrt_render would time the two against each other stay out of the per-kernel sums
export MRT_CAMERA_PACKETS=${3:-0}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --workload $wl --no-secondary --no-strong-proxy --serial > $out/bench.log 2>&1 || { echo "trace run failed"; tail -5 $out/bench.log; exit 1; }
echo "trace done"
short="--workload $wl --no-secondary --no-strong-proxy --serial --steps 1 --warmup 0 --no-cpu"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 400 rocprofv3 --pmc $c --output-format csv -d $out/$c -- python3 bench.py $short > $out/$c.log 2>&1 || { echo "pass $c failed"; tail -5 $out/$c.log; exit 1; }
  echo "pass $c done"
done
timeout -s KILL 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $out/sq -- python3 bench.py $short > $out/sq.log 2>&1 || { echo "pass sq failed"; exit 1; }
echo "pass sq done"
timeout -s KILL 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $out/tcc -- python3 bench.py $short > $out/tcc.log 2>&1 || { echo "pass tcc failed"; exit 1; }
echo "pass tcc done"
# the L2's read requests to the fabric by size (a miss of a 16-byte gather is one 128-byte request: profiles/r03_gather_calibration.json)
timeout -s KILL 400 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $out/ea -- python3 bench.py $short > $out/ea.log 2>&1 || { echo "pass ea failed"; exit 1; }
echo "pass ea done"
timeout -s KILL 400 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $out/tcp -- python3 bench.py $short > $out/tcp.log 2>&1 || { echo "pass tcp failed"; exit 1; }
echo all done
