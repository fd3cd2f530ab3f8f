#!/usr/bin/env python3
"""Metagenome worker throughput on one GPU (not the headline bench): zymo10-like synthetic community (8 circular bacterial genomes
of 1.9-6.8 Mb, 2 yeasts of 12 / 19 Mb in 16 linear chromosomes), hg002-like model, one ns_generate call per batch.
Prints reads/s and where the time goes (device kernels vs the host-side assign_species)."""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nanosim_amd import engine as E, metagenome as MG, model as M, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=1_000_000)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--chimeric", action="store_true")
a = ap.parse_args()
SEED = 20260926
sizes = [1.9e6, 2.7e6, 2.9e6, 3.0e6, 4.0e6, 4.8e6, 4.8e6, 6.8e6]
names, chunks, circ, off, species, keys = [], [], [], [0], [], []
for i, n in enumerate(sizes):
    sp = "Bacterium_%d" % i
    species.append(sp); keys.append(["chr"])
    names.append(sp + "-chr"); chunks.append(synth.synth_sequence(int(n), SEED + i, n_frac=0.0005, iupac_frac=0.0002, lower_frac=0.02, hp_boost=0.005))
    circ.append(1); off.append(off[-1] + 1)
for i, tot in enumerate((12e6, 19e6)):
    sp = "Yeast_%d" % i
    species.append(sp); keys.append(["chr%d" % c for c in range(16)])
    for c in range(16):
        names.append(sp + "-chr%d" % c); chunks.append(synth.synth_sequence(int(tot / 16), SEED + 100 * (i + 1) + c, n_frac=0.0005, iupac_frac=0.0002, lower_frac=0.02, hp_boost=0.005))
        circ.append(0)
    off.append(off[-1] + 16)
lens = np.array([len(c) for c in chunks], dtype=np.uint64)
ref = M.Reference(names, np.concatenate(chunks), np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64), np.array(circ, dtype=np.uint8))
mref = MG.MetaReference(ref, species, np.array(off, dtype=np.uint32), keys)
abun = {sp: v for sp, v in zip(species, [12, 12, 12, 12, 12, 12, 12, 12, 2, 2])}
tmp = tempfile.mkdtemp(prefix="nsmeta_")
prefix = os.path.join(tmp, "hg002_like")
synth.write_model(prefix, synth.SynthModelSpec(n_train=1_000_000, seed=SEED), write_pkl=False)
mdl = M.load_model(prefix, chimeric=a.chimeric)
infl = {sp: MG.inflate_abun(abun, sp, mdl.abun_inflation) for sp in abun} if a.chimeric else None
eng = E.Engine(0)
eng.set_metagenome(mref, abun, infl)
eng.load_model(mdl)
def step(i):
    return eng.generate(E.make_params(seed=SEED, first_read=i * a.reads, n_reads=a.reads, chimeric=a.chimeric, max_len=mref.max_chrom, meta=True))
step(0)
t0 = time.perf_counter()
infos = [step(1 + i).info for i in range(a.steps)]
dt = (time.perf_counter() - t0) / a.steps
print(json.dumps({"metric": "metagenome reads/s (one worker per batch)", "value": a.reads / dt, "ms_per_batch": dt * 1e3,
                  "device_ms": float(np.mean([x.ms_total for x in infos])),
                  "kernel_ms": {n: float(np.mean([x.ms_kernel[k] for x in infos])) for k, n in enumerate(E.KERNEL_NAMES)},
                  "passes": int(max(1 + int(b) for b in [0])), "reads": a.reads, "chimeric": bool(a.chimeric),
                  "species_bases_frac": (eng.species_bases() / eng.species_bases().sum()).round(4).tolist()}))
eng.close()
