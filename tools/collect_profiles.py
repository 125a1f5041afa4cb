#!/usr/bin/env python3
"""Copies the judged summaries of a tools/profile_bench.sh run into profiles/.

usage: tools/collect_profiles.py gpurun_out/prof_r02 r02 million_bunny

Writes profiles/<tag>_<workload>_kernel_stats.csv  rocprofv3 --kernel-trace --stats of `python3 bench.py --workload W --no-secondary`
       profiles/<tag>_<workload>_bench.json        the JSON line that run printed
       profiles/<tag>_<workload>_pmc_summary.json  per-kernel counter sums of the --pmc passes (one step)
       profiles/hbm_traffic.json                   per workload and kernel: HBM bytes per launch,
                                                   (2*FETCH_SIZE + WRITE_SIZE) KiB / dispatches (FETCH_SIZE
                                                   doubled: gfx950 tallies 128-byte read requests at 64
                                                   bytes), and the L2 (TCC) hit rate; read by bench.py
"""
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, tag, workload = sys.argv[1], sys.argv[2], sys.argv[3]
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
base = "%s_%s" % (tag, workload)

# (the bench also runs raytracer_amd/gather_bench as a child process, which the profiler traces as well: the
# file wanted is the one with the renderer's kernels)
stats = [f for f in glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")) if "extend_kernel" in open(f).read()][0]
shutil.copy(stats, os.path.join(dst, base + "_kernel_stats.csv"))
line = [l for l in open(os.path.join(src, "bench.log")) if l.startswith('{"metric"')][-1]
with open(os.path.join(dst, base + "_bench.json"), "w") as f:
    f.write(line)
tmp = os.path.join(dst, base + "_pmc_summary.json")
subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), src, "--json", tmp], check=True, stdout=subprocess.DEVNULL)
pmc = json.load(open(tmp))

path = os.path.join(dst, "hbm_traffic.json")
allw = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_* (separate passes) over `python3 bench.py --workload W --no-secondary "
                  "--steps 1 --warmup 0 --no-cpu`",
        "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes summed over the launches of the step / number of launches",
        "workloads": {}}
if os.path.exists(path):
    try:
        old = json.load(open(path))
        if "workloads" in old:
            allw["workloads"] = old["workloads"]
    except Exception:
        pass
w = {"tag": tag, "bytes_per_launch": {}, "bytes_per_step": {}, "fetch_size_kib_per_step": {}, "write_size_kib_per_step": {},
     "dispatches_per_step": {}, "tcc_hit_rate": {}, "l2_requests_per_step": {}, "l2_misses_per_step": {},
     "fabric_read_requests_per_step": {}, "fabric_read_bytes_per_step": {}, "l1_to_l2_read_requests_per_step": {}}
for k, c in pmc.items():
    if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        continue
    b = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
    w["bytes_per_step"][k] = b
    w["bytes_per_launch"][k] = b / c["dispatches"]
    w["dispatches_per_step"][k] = c["dispatches"]
    w["fetch_size_kib_per_step"][k] = c["FETCH_SIZE"]
    w["write_size_kib_per_step"][k] = c["WRITE_SIZE"]
    if c.get("TCC_REQ_sum"):
        w["tcc_hit_rate"][k] = c["TCC_HIT_sum"] / c["TCC_REQ_sum"]
        w["l2_requests_per_step"][k] = c["TCC_REQ_sum"]
        w["l2_misses_per_step"][k] = c["TCC_MISS_sum"]
    if c.get("TCC_EA0_RDREQ_sum"):
        # the L2's read requests to the fabric, and their bytes by request size
        w["fabric_read_requests_per_step"][k] = c["TCC_EA0_RDREQ_sum"]
        w["fabric_read_bytes_per_step"][k] = 32.0 * c.get("TCC_EA0_RDREQ_32B_sum", 0) + 64.0 * c.get("TCC_EA0_RDREQ_64B_sum", 0) + 128.0 * c.get("TCC_EA0_RDREQ_128B_sum", 0)
    if c.get("TCP_TCC_READ_REQ_sum"):
        w["l1_to_l2_read_requests_per_step"][k] = c["TCP_TCC_READ_REQ_sum"]
allw["workloads"][workload] = w
with open(path, "w") as f:
    json.dump(allw, f, indent=1)
# The bench line of the trace run read the counter traffic from the hbm_traffic.json of the run
# before; bring its copy in profiles/ in step with the passes just collected.
bench_path = os.path.join(dst, base + "_bench.json")
line = json.loads(open(bench_path).read())
roof = line.get("roofline") or {}
kernel = (roof.get("kernel") or "").split(" ")[0]
if kernel in w["bytes_per_launch"] and roof.get("avg_launch_ms"):
    roof["traffic"] = w["bytes_per_launch"][kernel]
    roof["hbm_frac_from_counters"] = roof["traffic"] / (roof["avg_launch_ms"] * 1e-3) / 1e9 / roof["peak"]
    if kernel in w["tcc_hit_rate"]:
        roof["limiter"] = "fabric line fetches: every L2 miss of a 16-byte gather moves a 128-byte line (TCC hit rate %.2f)" % w["tcc_hit_rate"][kernel]
    with open(bench_path, "w") as f:
        f.write(json.dumps(line) + "\n")
print("profiles written for", tag, workload)
