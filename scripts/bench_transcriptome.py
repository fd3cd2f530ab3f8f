#!/usr/bin/env python3
"""Transcriptome worker throughput on one GPU (not the headline bench): synthetic transcriptome (100 000 transcripts, median 1.4 kb,
log-normal expression, 60 % with polyA), hg002-like error model with the 2-D length KDE, one ns_generate call per batch."""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nanosim_amd import engine as E, intron_retention as IR, model as M, synth, transcriptome as T  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=2_000_000)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--transcripts", type=int, default=100_000)
ap.add_argument("--fastq", action="store_true")
ap.add_argument("--model-ir", action="store_true", help="intron retention: synthetic exon/intron structure on one chromosome, P(IR | start) = 0.1")
a = ap.parse_args()
SEED = 20260926
rng = np.random.Generator(np.random.Philox(SEED))
n = a.transcripts
lens = np.clip(np.rint(rng.lognormal(np.log(1400.0), 0.6, n)), 150, 30000).astype(np.int64)
bases = synth.synth_sequence(int(lens.sum()), SEED, iupac_frac=0.0002)
off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
ref = M.Reference(["ENST%011d" % (i + 1) for i in range(n)], bases, off, np.zeros(n, dtype=np.uint8))
tpm = rng.lognormal(1.0, 2.0, n)
tpm[rng.random(n) < 0.2] = 0.0
dict_exp = {ref.names[i]: float(tpm[i]) for i in range(n) if tpm[i] > 0}
names, weights = T.make_cdf(dict_exp, {ref.names[i]: int(lens[i]) for i in range(n)})
index = {nm: i for i, nm in enumerate(ref.names)}
tr = T.TranscriptomeReference(ref, np.array([index[k] for k, _ in names], dtype=np.uint32), np.cumsum(weights), np.array(weights),
                              (rng.random(n) < 0.6).astype(np.uint8), T.POLYA_SCALE_DEFAULT)
tmp = tempfile.mkdtemp(prefix="nstrx_")
prefix = os.path.join(tmp, "hg002_like")
synth.write_model(prefix, synth.SynthModelSpec(n_train=1_000_000, seed=SEED), write_pkl=False)
mdl = M.load_model(prefix, transcriptome=True, fastq=a.fastq)
eng = E.Engine(0)
eng.set_transcriptome(tr)
if a.model_ir:        # every transcript: 1-8 exons that add up to its length, introns of 100-2000 bases between them, '+' strand
    n_ex = np.minimum(rng.integers(1, 9, n), np.maximum(lens // 100, 1))
    ty, st, ln, item_off, pos = [], [], [], np.zeros(n + 1, dtype=np.uint32), 1000
    for i in range(n):
        cuts = np.sort(rng.choice(np.arange(1, lens[i]), size=int(n_ex[i]) - 1, replace=False)) if n_ex[i] > 1 else np.zeros(0, dtype=np.int64)
        ex = np.diff(np.concatenate([[0], cuts, [lens[i]]]))
        for j, e in enumerate(ex):
            ty.append(0); st.append(pos); ln.append(int(e)); pos += int(e)
            if j + 1 < len(ex):
                il = int(rng.integers(100, 2000)); ty.append(1); st.append(pos); ln.append(il); pos += il
        pos += 200
        item_off[i + 1] = len(ty)
    genome = M.Reference(["1"], synth.synth_sequence(pos + 1000, SEED + 1, lower_frac=0.3), np.array([0, pos + 1000], dtype=np.uint64), np.zeros(1, dtype=np.uint8))
    ir = IR.IntronRetention(genome=genome, item_off=item_off, item_type=np.array(ty, dtype=np.uint8), item_minus=np.zeros(len(ty), dtype=np.uint8),
                            item_chrom=np.zeros(len(ty), dtype=np.uint32), item_start=np.array(st, dtype=np.uint32), item_len=np.array(ln, dtype=np.uint32),
                            p_no_ir=[0.9, 0.95, 0.7], p_ir=[0.1, 0.05, 0.3], eligible=np.ones(n, dtype=bool))
    eng.set_intron_retention(ir)
eng.load_model(mdl)
def step(i):
    return eng.generate(E.make_params(seed=SEED, first_read=i * a.reads, n_reads=a.reads, max_len=ref.max_chrom, trx=True, fastq=a.fastq, model_ir=a.model_ir))
step(0)
t0 = time.perf_counter()
infos = [step(1 + i).info for i in range(a.steps)]
dt = (time.perf_counter() - t0) / a.steps
print(json.dumps({"metric": "transcriptome reads/s (aligned reads, one worker call per batch)", "value": a.reads / dt, "ms_per_batch": dt * 1e3,
                  "bases_per_s": float(np.mean([int(x.total_bases) for x in infos])) / dt, "mean_read_len": float(np.mean([int(x.total_bases) for x in infos])) / a.reads,
                  "kernel_ms": {nm: float(np.mean([x.ms_kernel[k] for x in infos])) for k, nm in enumerate(E.KERNEL_NAMES)},
                  "reads": a.reads, "transcripts": n, "fastq": bool(a.fastq), "model_ir": bool(a.model_ir), "spliced_bytes": [int(x.spliced_bytes) for x in infos], "replans": [int(x.n_overflow) for x in infos], "device_ms": float(np.mean([x.ms_total for x in infos]))}))
eng.close()
