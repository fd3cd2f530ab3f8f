#!/usr/bin/env python3
"""End-to-end CLI throughput (device + D2H + file writes), genome mode: E. coli-like reference, hg002-like model, files on /dev/shm."""
import argparse
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nanosim_amd import simulator, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("-n", type=int, default=2_000_000)
ap.add_argument("--fastq", action="store_true")
ap.add_argument("--dir", default="/dev/shm")
ap.add_argument("-t", type=int, default=1, help="sub-files per batch (the CLI's -t)")
ap.add_argument("--no-merge", action="store_true", help="keep the sub-files (the CLI's --no-merge)")
a = ap.parse_args()
d = tempfile.mkdtemp(prefix="nscli_", dir=a.dir)
try:
    prefix = os.path.join(d, "training")
    synth.write_model(prefix, synth.SynthModelSpec(n_train=1_000_000, seed=1), write_pkl=False)
    fa = os.path.join(d, "ecoli_like.fa")
    synth.write_fasta(fa, [("ecoli-like", synth.synth_sequence(synth.ECOLI_LEN, 1, n_frac=0.0005, iupac_frac=0.0002, lower_frac=0.02, hp_boost=0.005))])
    out = os.path.join(d, "sim")
    argv = ["genome", "-rg", fa, "-c", prefix, "-o", out, "-n", str(a.n), "--seed", "1", "-dna_type", "circular", "-t", str(a.t)] + (["--fastq"] if a.fastq else []) + (["--no-merge"] if a.no_merge else [])
    so = sys.stdout
    sys.stdout = open(os.devnull, "w")
    t0 = time.perf_counter()
    try:
        simulator.main(argv)
    finally:
        sys.stdout = so
    dt = time.perf_counter() - t0
    size = sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d) if f.startswith("sim_"))
    print("CLI end to end: %d reads in %.2f s = %.2f M reads/s, %.2f GB/s of output files (incl. model/reference loading)" % (a.n, dt, a.n / dt / 1e6, size / dt / 1e9))
finally:
    shutil.rmtree(d, ignore_errors=True)
