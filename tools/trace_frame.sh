#!/bin/bash
# Kernel trace of one frame of the strong-split proxy (tools/strong_proxy.py --only SPP) into gpurun_out/<name>/.
# usage (on the GPU box): tools/trace_frame.sh <name> <spp> [extra strong_proxy.py args]
name=$1; spp=$2; shift 2
out=$GRAFT_REPO_ROOT/gpurun_out/$name
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace -d $out -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/strong_proxy.py --only $spp --steps 1 "$@" > $out.log 2>&1
