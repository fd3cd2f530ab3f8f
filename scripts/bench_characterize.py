#!/usr/bin/env python3
"""Training-side histogramming throughput on one GPU (not the headline bench): cs strings of N synthetic alignments (built from the event
lists of reads the engine itself generates with the hg002-like model, ~1 KB of cs per 8 kb read) through ns_cs_histograms; prints
alignments/s, the kernel's own time and its fraction of the HBM roofline (algorithmic bytes = the cs strings, read once).
    python scripts/bench_characterize.py [--alignments 200000]"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nanosim_amd import characterize, engine as E, model as M, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--alignments", type=int, default=200_000)
ap.add_argument("--steps", type=int, default=3)
a = ap.parse_args()
SEED = 20260926
tmp = tempfile.mkdtemp(prefix="nschar_")
prefix = os.path.join(tmp, "hg002_like")
synth.write_model(prefix, synth.SynthModelSpec(n_train=200_000, seed=SEED), write_pkl=False)
mdl = M.load_model(prefix)
seq = synth.synth_sequence(synth.ECOLI_LEN, SEED, n_frac=0.0005, iupac_frac=0.0002, lower_frac=0.02, hp_boost=0.005)
ref = M.Reference(["ecoli-like"], seq, np.array([0, len(seq)], dtype=np.uint64), np.array([1], dtype=np.uint8))
eng = E.Engine(0)
eng.set_reference(ref)
eng.load_model(mdl)
n_src = min(a.alignments, 20_000)                       # cs strings of 20 000 reads, repeated up to the requested number
b = eng.generate(E.make_params(seed=SEED, first_read=0, n_reads=n_src, max_len=ref.max_chrom, emit_records=False))
pieces, events = b.pieces(), b.events()
rng = np.random.default_rng(SEED)
letters = np.frombuffer(b"acgt", dtype=np.uint8)
cs = []
for pc in pieces:
    ev = events[int(pc["ev_off"]):int(pc["ev_off"]) + int(pc["n_ev"])]
    s, pos = [], 0
    for e in ev:
        epos, ln, ty = int(e["pos"]), int(e["info"]) & 0xfff, (int(e["info"]) >> 12) & 3
        if epos > pos:
            s.append(":%d" % (epos - pos)); pos = epos
        if ty == 0:
            s.append("*ac" * ln); pos += ln
        elif ty == 1:
            s.append("+" + "a" * ln)
        else:
            s.append("-" + "a" * ln); pos += ln
    if int(pc["ref_len"]) > pos:
        s.append(":%d" % (int(pc["ref_len"]) - pos))
    cs.append("".join(s))
cs = (cs * (a.alignments // len(cs) + 1))[:a.alignments]
nbytes = sum(len(x) for x in cs)
t = characterize.count(eng, cs)                          # sizes the match matrix, warms up
t0 = time.perf_counter()
ms = []
for _ in range(a.steps):
    ms.append(characterize.count(eng, cs, cap=t["match_list"].shape[0])["ms_kernel"])
dt = (time.perf_counter() - t0) / a.steps
print(json.dumps({"metric": "training-side histogramming, alignments/s (ns_cs_histograms incl. packing + H2D)", "value": a.alignments / dt,
                  "alignments": a.alignments, "cs_bytes": nbytes, "cs_bytes_per_alignment": nbytes / a.alignments,
                  "kernel_ms": float(np.mean(ms)), "kernel_gb_per_s": nbytes / (float(np.mean(ms)) * 1e-3) / 1e9,
                  "kernel_frac_of_hbm_8tbs": nbytes / (float(np.mean(ms)) * 1e-3) / 1e9 / 8000.0, "max_match": t["max_match"]}))
eng.close()
