#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (one directory per pass) into per-kernel means per launch."""
import csv, glob, os, sys, collections, json
root = sys.argv[1]
out = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            if not k.startswith("bsr::"):
                continue
            out[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(root, "*", "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            if k.startswith("bsr::"):
                dur[k].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
res = {}
for k in sorted(out):
    r = {c: sum(v) / len(v) for c, v in out[k].items()}
    if dur[k]:
        r["dur_us(profiled)"] = sum(dur[k]) / len(dur[k])
    if "FETCH_SIZE" in r:
        r["hbm_read_MB(2x FETCH_SIZE KiB, gfx950 wide-read correction)"] = 2 * r["FETCH_SIZE"] * 1024 / 1e6
        r["hbm_read_MB(raw)"] = r["FETCH_SIZE"] * 1024 / 1e6
    if "WRITE_SIZE" in r:
        r["hbm_write_MB"] = r["WRITE_SIZE"] * 1024 / 1e6
    res[k] = r
print(json.dumps(res, indent=1))
