#!/usr/bin/env python3
"""Folds the rocprofv3 passes of scripts/profile_round.sh (gpurun_out/<tag>/) into profiles/<tag>/: per configuration the bench
lines, kernel_stats_<cfg>.csv (per-kernel average durations of the bench command) and pmc_<cfg>.json — per kernel the mean counter
value per launch and HBM bytes per launch / per read as MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE are in KB;
FETCH_SIZE is doubled on gfx950)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
reads = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", tag)
dst = os.path.join(root, "profiles", tag)
os.makedirs(dst, exist_ok=True)
for cfg_dir in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    cfg = os.path.basename(cfg_dir)[4:]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(cfg_dir, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"].replace("void ", "").split("(")[0]
            if name.startswith("k_"):
                acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    kernels = {}
    for name, cs in sorted(acc.items()):
        d = {c: sum(v) / len(v) for c, v in cs.items()}
        d["launches_averaged"] = max(len(v) for v in cs.values())
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            d["hbm_bytes_per_launch"] = (2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024
            d["hbm_bytes_per_read"] = d["hbm_bytes_per_launch"] / reads
        for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
            if c in d:
                d[c.replace("SQ_INSTS_", "").lower() + "_per_read"] = d[c] / reads
        if "SQ_LDS_BANK_CONFLICT" in d and d.get("SQ_LDS_IDX_ACTIVE"):
            d["lds_conflict_share"] = d["SQ_LDS_BANK_CONFLICT"] / d["SQ_LDS_IDX_ACTIVE"]
        kernels[name] = d
    out = {"note": "rocprofv3 PMC, separate passes (scripts/profile_round.sh), bench.py --reads %d --aligned-only; values per launch (mean "
                   "over the warm-up and the timed launch). hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) KB: FETCH_SIZE is doubled per "
                   "MI355X_MICROARCH.md (gfx950 counts 128-B requests at 64 B for wide coalesced reads); WRITE_SIZE is uncalibrated for "
                   "partial-line writes." % reads,
           "config": cfg, "reads_per_launch": reads, "kernels": kernels}
    json.dump(out, open(os.path.join(dst, "pmc_%s.json" % cfg), "w"), indent=1)
    print(cfg, {k: round(v["hbm_bytes_per_read"]) for k, v in kernels.items() if "hbm_bytes_per_read" in v and v["hbm_bytes_per_read"] > 100})
    for f in glob.glob(os.path.join(src, "stats_%s" % cfg, "**", "*kernel_stats.csv"), recursive=True):
        rows = list(csv.reader(open(f)))
        with open(os.path.join(dst, "kernel_stats_%s.csv" % cfg), "w", newline="") as g:
            w = csv.writer(g)
            for r in rows:                          # (the rocPRIM template names run to kilobytes)
                w.writerow([r[0][:120]] + r[1:])
    for nm in ("bench_%s.log" % cfg, "bench_%s_under_rocprof.log" % cfg):
        p = os.path.join(src, nm)
        if os.path.exists(p):
            lines = [ln for ln in open(p) if ln.startswith("{")]
            if lines:
                open(os.path.join(dst, nm.replace(".log", ".json")), "w").write(lines[-1])
