#!/usr/bin/env python3
"""Chimeric batches of many sizes and seeds on the GPU: a batch large enough for the wave-per-read lists and the thread-per-piece chain
(nanosim_amd.hip: visiting_order, k_chain's piece modes) must be the same bytes as the same reads generated in batches of 4 096 (one
thread per read: the path the oracle parity tests cover) — records, error profile, per-read lengths.  Sizes around every threshold of the
host logic (coop_min = 16 384; few reads of several pieces: all of them on the wave-per-read list; segment means 1.05 and 2.5).
    python scripts/parity_chimeric_big.py"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nanosim_amd import engine as E, model as M, synth  # noqa: E402

SEED = 20260926
bad = 0
# the table shape of a trained model (15 previous-match bins x 1 500-row ECDFs: the LDS image holds the hot prefixes only, 512-thread workgroups)
TRAINED = dict(ecdf_rows=1500, mm_bins=((0, 1), (1, 2), (2, 3), (3, 5), (5, 7), (7, 10), (10, 14), (14, 19), (19, 25), (25, 33), (33, 45), (45, 60), (60, 90), (90, 150), (150, 1500)),
               mm_means=(24.0, 25.0, 26.0, 27.0, 28.0, 29.0, 30.0, 31.0, 31.0, 32.0, 33.0, 34.0, 35.0, 36.0, 36.0), mm_zero=(0.0,) + (0.03,) * 14)
for seg_mean in (1.05, 2.5, 1.002, -1.05):                      # (negative: the trained table shape)
    prefix = os.path.join(tempfile.mkdtemp(prefix="nschim_"), "training")
    synth.write_model(prefix, synth.SynthModelSpec(n_train=200_000, seed=SEED, segment_mean=abs(seg_mean), **(TRAINED if seg_mean < 0 else {})), write_pkl=False)
    mdl = M.load_model(prefix, fastq=True, homopolymer=True, chimeric=True)
    bases = synth.synth_sequence(synth.ECOLI_LEN, SEED, n_frac=0.0005, iupac_frac=0.0002, lower_frac=0.02, hp_boost=0.005)
    for circ in (1, 0):
        ref = M.Reference(["ecoli-like"], bases, np.array([0, synth.ECOLI_LEN], dtype=np.uint64), np.array([circ], dtype=np.uint8))
        eng = E.Engine(0); eng.set_reference(ref); eng.load_model(mdl)
        cases = ((16_384, dict(emit_errlog=True)), (17_000, dict(fastq=True)), (40_000, dict(emit_errlog=True, fastq=True, kmer_bias=5)),
                 (70_000, dict(emit_errlog=True)), (150_000, dict()))
        if os.environ.get("NS_PCB_CASES"):                       # e.g. NS_PCB_CASES=3,4: a subset (debugging)
            cases = tuple(cases[int(i)] for i in os.environ["NS_PCB_CASES"].split(","))
        for n, extra in cases:
            for seed in (1, 2):
                kw = dict(seed=SEED + 100 * seed, chimeric=True, max_len=ref.max_chrom, **extra)
                b = eng.generate(E.make_params(first_read=7 * seed, n_reads=n, **kw))
                reads = b.reads()
                multi = float((reads["n_pieces"] > 1).mean())
                rec = b.records().copy(); err = b.errlog().copy() if extra.get("emit_errlog") else np.zeros(0, np.uint8)
                sl = reads["seq_len"].copy(); ok = bool(np.all(reads["flags"] == 0))
                r_at = e_at = 0
                for f in range(0, n, 4096):
                    m = min(4096, n - f)
                    c = eng.generate(E.make_params(first_read=7 * seed + f, n_reads=m, **kw))
                    cr = c.records(); ce = c.errlog() if extra.get("emit_errlog") else np.zeros(0, np.uint8)
                    ok = ok and np.array_equal(c.reads()["seq_len"], sl[f:f + m]) and np.array_equal(cr, rec[r_at:r_at + len(cr)]) and np.array_equal(ce, err[e_at:e_at + len(ce)])
                    r_at += len(cr); e_at += len(ce)
                ok = ok and r_at == len(rec) and e_at == len(err)
                bad += 0 if ok else 1
                print("segment mean %-5s %-8s n %6d seed %d %-52s share of reads of several pieces %.3f  %s" % (seg_mean, "circular" if circ else "linear", n, seed, extra, multi, "identical" if ok else "DIFFERENT"), flush=True)
        eng.close()
print("chimeric big-batch parity:", "all identical" if not bad else "%d cases DIFFER" % bad)
sys.exit(1 if bad else 0)
