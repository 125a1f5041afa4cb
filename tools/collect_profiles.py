#!/usr/bin/env python3
"""Copies the judged summaries of a tools/profile_bench.sh run into profiles/.

usage: tools/collect_profiles.py gpurun_out/prof_r01 r01

Writes profiles/<tag>_bench_kernel_stats.csv  rocprofv3 --kernel-trace --stats of `python3 bench.py`
       profiles/<tag>_bench.json              the JSON line that run printed
       profiles/<tag>_pmc_summary.json        per-kernel counter sums of the --pmc passes (one step)
       profiles/hbm_traffic.json              bytes per step per kernel, (2*FETCH_SIZE + WRITE_SIZE) KiB:
                                              FETCH_SIZE is doubled as gfx950 tallies 128-byte read
                                              requests at 64 bytes; read by bench.py for roofline.traffic
"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, tag = sys.argv[1], sys.argv[2]
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))[0]
shutil.copy(stats, os.path.join(dst, tag + "_bench_kernel_stats.csv"))
line = [l for l in open(os.path.join(src, "bench.log")) if l.startswith('{"metric"')][-1]
with open(os.path.join(dst, tag + "_bench.json"), "w") as f:
    f.write(line)
tmp = os.path.join(dst, tag + "_pmc_summary.json")
subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), src, "--json", tmp], check=True, stdout=subprocess.DEVNULL)
pmc = json.load(open(tmp))


def traffic(k):
    return (2.0 * pmc[k]["FETCH_SIZE"] + pmc[k]["WRITE_SIZE"]) * 1024.0


out = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `python3 bench.py --steps 1 --warmup 0 --no-cpu`",
       "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes, summed over the launches of one step",
       "bytes_per_step": {k: traffic(k) for k in pmc if "FETCH_SIZE" in pmc[k]}}
for mis in ("connect_mis_occlusion_kernel", "connect_mis_kernel"):  # whichever the scene made the library use
    if mis in pmc:
        out["bytes_per_step"]["connect_shadow_kernel+" + mis] = traffic("connect_shadow_kernel") + traffic(mis)
out["fetch_size_kib_per_step"] = {k: pmc[k]["FETCH_SIZE"] for k in pmc if "FETCH_SIZE" in pmc[k]}
out["write_size_kib_per_step"] = {k: pmc[k]["WRITE_SIZE"] for k in pmc if "WRITE_SIZE" in pmc[k]}
with open(os.path.join(dst, "hbm_traffic.json"), "w") as f:
    json.dump(out, f, indent=1)
print("profiles written for", tag)
