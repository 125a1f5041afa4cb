#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter_collection.csv files per kernel and counter.

usage: tools/pmc_summary.py <dir with pass subdirectories> [--json out.json]
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    name = name.split("(")[0]
    if "relay_trace_kernel" in name or "relay_finish_kernel" in name:  # the follow-up launches of the long rays
        return "relay_kernels"
    if "extend_camera" in name:  # the first-bounce instances of the extend traversal (streaming or packets)
        return "extend_kernel"
    for k in ("extend_batch_kernel", "extend_kernel", "connect_shadow_kernel", "connect_mis_occlusion_kernel", "connect_mis_kernel", "mis_resolve_kernel", "shade_kernel",
              "generate_kernel", "accumulate_kernel", "resolve_kernel"):
        if k in name:
            return k
    return None


def main():
    root = sys.argv[1]
    tot = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(lambda: defaultdict(int))
    for f in sorted(glob.glob(os.path.join(root, "*", "*", "*_counter_collection.csv"))):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = short(row["Kernel_Name"])
                if k is None:
                    continue
                tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
                calls[k][row["Counter_Name"]] += 1
    out = {}
    for k in sorted(tot):
        out[k] = {c: tot[k][c] for c in sorted(tot[k])}
        out[k]["dispatches"] = max(calls[k].values())
        print(k, "dispatches", out[k]["dispatches"])
        for c in sorted(tot[k]):
            print("   %-34s %18.0f" % (c, tot[k][c]))
    if "--json" in sys.argv:
        with open(sys.argv[sys.argv.index("--json") + 1], "w") as fh:
            json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
