#!/bin/bash
# Per-kernel device time of a few serial frames of one workload (kernel trace + stats).
# usage: tools/quick_stats.sh <outdir> [workload] [camera rays as packets: 0 | 1]
out=$1; wl=${2:-million_bunny}
export MRT_CAMERA_PACKETS=${3:-0}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --workload $wl --no-secondary --no-strong-proxy --serial --steps 3 --warmup 1 --no-cpu > $out/bench.log 2>&1 || { echo "trace run failed"; tail -5 $out/bench.log; exit 1; }
f=$(ls $out/trace/*/*kernel_stats.csv | head -1)
python3 - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print('%-60s calls %6s  total %9.2f ms  avg %8.3f ms  %5s %%' % (r['Name'][:60], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e6, r['Percentage']))
PY
