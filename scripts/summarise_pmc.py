#!/usr/bin/env python3
"""Folds the rocprofv3 PMC passes of scripts/profile_round.sh (gpurun_out/<tag>/pmc_*) into profiles/<tag>/pmc_summary.json:
per kernel the mean counter value per launch, and HBM bytes per launch / per read as MI355X_MICROARCH.md prescribes
(FETCH_SIZE and WRITE_SIZE are in KB; FETCH_SIZE is doubled on gfx950)."""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
reads = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", tag)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].replace("void ", "").split("(")[0]
        if not name.startswith("k_"):
            continue
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
kernels = {}
for name, cs in sorted(acc.items()):
    d = {}
    for c, v in cs.items():
        d[c] = sum(v) / len(v)
    d["launches_averaged"] = max(len(v) for v in cs.values())
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        d["hbm_bytes_per_launch"] = (2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024
        d["hbm_bytes_per_read"] = d["hbm_bytes_per_launch"] / reads
    kernels[name] = d
out = {"note": "rocprofv3 PMC, separate passes (scripts/profile_round.sh), bench.py --reads %d; values per launch (mean over the "
               "warm-up and the timed launch). hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) KB: FETCH_SIZE is doubled per "
               "MI355X_MICROARCH.md (gfx950 counts 128-B requests at 64 B for wide coalesced reads); WRITE_SIZE is "
               "uncalibrated for partial-line writes." % reads,
       "reads_per_launch": reads, "kernels": kernels}
os.makedirs(os.path.join(root, "profiles", tag), exist_ok=True)
json.dump(out, open(os.path.join(root, "profiles", tag, "pmc_summary.json"), "w"), indent=1)
for k in ("k_words", "k_materialise<false>", "k_chain<true, false>"):
    if k in kernels and "hbm_bytes_per_read" in kernels[k]:
        print(k, "hbm bytes/read", round(kernels[k]["hbm_bytes_per_read"]))
