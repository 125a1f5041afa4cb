#!/bin/bash
# L2 misses and fabric read requests per kernel of one frame, for library variants and environment settings:
# usage: tools/pmc_misses.sh <outdir> <workload> <variant .so name under raytracer_amd/ab | -> [VAR=value ...]
out=$1; wl=$2; so=$3; shift 3
for kv in "$@"; do export "$kv"; done
export MRT_CAMERA_PACKETS=0
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
[ "$so" != "-" ] && cp raytracer_amd/ab/$so.so raytracer_amd/libmrt_hip.so
mkdir -p $out
timeout -s KILL 400 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc -- python3 bench.py --workload $wl --no-secondary --no-strong-proxy --serial --steps 1 --warmup 0 --no-cpu > $out/run.log 2>&1 || { echo "pass failed"; tail -5 $out/run.log; exit 1; }
python3 - $out <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + '/pmc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].split('<')[0].replace('void ', '')
        tot[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, v in sorted(tot.items(), key=lambda kv: -kv[1].get('TCC_MISS_sum', 0))[:8]:
    print('%-34s miss %8.3f G  hit %8.3f G  ea_rd %8.3f G' % (k, v['TCC_MISS_sum'] / 1e9, v['TCC_HIT_sum'] / 1e9, v['TCC_EA0_RDREQ_sum'] / 1e9))
PY
