#!/bin/bash
# Profiles of the default benchmark command for profiles/: kernel trace + stats, then one
# hardware-counter pass per rocprofv3 run (FETCH_SIZE and WRITE_SIZE do not fit one pass).
# usage: tools/profile_bench.sh <outdir under gpurun_out>
out=${1:-gpurun_out/prof_bench}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py > $out/bench.log 2>&1 || { echo "trace run failed"; tail -5 $out/bench.log; exit 1; }
echo "trace done"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --pmc $c --output-format csv -d $out/$c -- python3 bench.py --steps 1 --warmup 0 --no-cpu > $out/$c.log 2>&1 || { echo "pass $c failed"; tail -5 $out/$c.log; exit 1; }
  echo "pass $c done"
done
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $out/sq -- python3 bench.py --steps 1 --warmup 0 --no-cpu > $out/sq.log 2>&1 || { echo "pass sq failed"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $out/tcc -- python3 bench.py --steps 1 --warmup 0 --no-cpu > $out/tcc.log 2>&1 || { echo "pass tcc failed"; exit 1; }
echo all done
