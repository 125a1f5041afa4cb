#!/usr/bin/env python3
"""Compiles the device code with -Rpass-analysis=kernel-resource-usage and prints, per kernel,
VGPRs, SGPRs, scratch bytes per lane, LDS bytes and occupancy (waves per SIMD)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import build  # noqa: E402

cmd = ["hipcc"] + [f for f in build.HIP_FLAGS if f not in ("-shared",)] + sys.argv[1:] + \
      ["-I" + build.INC, "-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null", os.path.join(build.CSRC, "mrt_hip.hip")]
txt = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
cur = None
rows = {}
for line in txt.splitlines():
    m = re.search(r"remark: Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+(?:\[[^\]]*\])?[A-Za-z /]*):\s*(\S+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = m.group(2)
for name, r in rows.items():
    short = subprocess.run(["c++filt", name], stdout=subprocess.PIPE, text=True).stdout.strip().split("(")[0]
    print("%-44s vgpr %3s agpr %3s sgpr %3s scratch %5s lds %6s occupancy %s" % (
        short[-44:], r.get("VGPRs"), r.get("AGPRs"), r.get("SGPRs"), r.get("ScratchSize [bytes/lane]"),
        r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]")))
if not rows:
    print(txt[-3000:])
