"""Summarises tools/gather_calibration.sh: per configuration of gather_bench, the counters per gathered
record (known: every record is one random 32-, 64- or 128-byte read inside one 128-byte line).
usage: gather_calibration_summary.py <dir> [--json out.json]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
out = {}
for cfg_dir in sorted(glob.glob(os.path.join(root, "*_*_*_*_*"))):
    if not os.path.isdir(cfg_dir):
        continue
    mode, mb, waves, rb, steps = os.path.basename(cfg_dir).split("_")
    mb, waves, rb, steps = int(mb), int(waves), int(rb), int(steps)
    lanes = 256 * waves * 256
    records = lanes * steps / (1 if rb == 32 else rb // 16)
    tot = defaultdict(float)
    n = defaultdict(int)
    for f in glob.glob(os.path.join(cfg_dir, "*", "*", "*_counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            if "gather" not in row["Kernel_Name"]:
                continue
            tot[row["Counter_Name"]] += float(row["Counter_Value"])
            n[row["Counter_Name"]] += 1
    per = {c: tot[c] / n[c] for c in tot}  # per dispatch
    r = {"records_per_dispatch": records, "record_bytes": rb, "table_mb": mb}
    for c, v in sorted(per.items()):
        r[c + "_per_record"] = v / records
    if "FETCH_SIZE" in per:
        r["FETCH_SIZE_bytes_per_record"] = per["FETCH_SIZE"] * 1024 / records
    if "TCC_EA0_RDREQ_sum" in per:
        b32, b64, b128 = per.get("TCC_EA0_RDREQ_32B_sum", 0), per.get("TCC_EA0_RDREQ_64B_sum", 0), per.get("TCC_EA0_RDREQ_128B_sum", 0)
        r["fabric_read_bytes_per_record_by_size"] = (32 * b32 + 64 * b64 + 128 * b128) / records
    out[os.path.basename(cfg_dir)] = r
    print(os.path.basename(cfg_dir))
    for k, v in r.items():
        print("   %-44s %14.4f" % (k, v))
if "--json" in sys.argv:
    json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
