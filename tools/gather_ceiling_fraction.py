#!/usr/bin/env python3
"""Re-computes the `kernels` section of profiles/r02_gather_ceiling.json from the committed
summaries of one workload: the traversal kernels' L1-miss read requests to L2 per second against the
rate a loop of nothing but dependent 32-byte gathers reaches at the same L2 hit rate.

usage: tools/gather_ceiling_fraction.py [workload]          (default million_bunny)

reads  profiles/r02_<workload>_mem_pipe.json       TCP_TCC_READ_REQ_sum per kernel class (one step)
       profiles/r02_<workload>_pmc_summary.json    TCC_HIT_sum / TCC_REQ_sum, launches per step
       profiles/r02_<workload>_kernel_stats.csv    rocprofv3 --stats: total duration and calls per kernel
       profiles/r02_gather_ceiling.json            R_hit / R_miss of raytracer_amd/gather_bench
"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
wl = sys.argv[1] if len(sys.argv) > 1 else "million_bunny"
mem = json.load(open(os.path.join(P, "r02_%s_mem_pipe.json" % wl)))
pmc = json.load(open(os.path.join(P, "r02_%s_pmc_summary.json" % wl)))
path = os.path.join(P, "r02_gather_ceiling.json")
g = json.load(open(path))
r_hit, r_miss = g["R_hit_G_per_s"], g["R_miss_G_per_s"]

# seconds per step of a kernel class: the timed instantiations (<false>: no visit counters) of the
# trace run, total duration / (calls / launches per step)
dur, calls = {}, {}
for row in csv.DictReader(open(os.path.join(P, "r02_%s_kernel_stats.csv" % wl))):
    name = row["Name"]
    if "<true>" in name.split("(")[0]:
        continue
    for k in ("connect_shadow_kernel", "connect_mis_occlusion_kernel", "extend_camera_kernel", "extend_kernel"):
        if k in name:
            k = "extend_kernel" if k == "extend_camera_kernel" else k
            dur[k] = dur.get(k, 0.0) + float(row["TotalDurationNs"]) * 1e-9
            calls[k] = calls.get(k, 0) + int(row["Calls"])
            break

out = {}
for k in ("extend_kernel", "connect_shadow_kernel", "connect_mis_occlusion_kernel"):
    steps = calls[k] / pmc[k]["dispatches"]
    sec = dur[k] / steps
    req = mem[k]["TCP_TCC_READ_REQ_sum"]
    h = pmc[k]["TCC_HIT_sum"] / pmc[k]["TCC_REQ_sum"]
    ceiling = 1.0 / (h / r_hit + (1.0 - h) / r_miss)
    out[k] = {"l2_read_requests_per_step": req, "seconds_per_step": sec, "g_requests_per_s": req / sec / 1e9, "l2_hit_rate": h,
              "ceiling_g_per_s": ceiling, "fraction": req / sec / 1e9 / ceiling}
    print("%-30s %.1f G requests/s  hit %.2f  ceiling %.1f  fraction %.2f" % (k, out[k]["g_requests_per_s"], h, ceiling, out[k]["fraction"]))
g["kernels"] = out
with open(path, "w") as f:
    json.dump(g, f, indent=1)
    f.write("\n")
