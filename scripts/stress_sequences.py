#!/usr/bin/env python3
"""Random sequences of worker calls on ONE long-lived engine (and its step companion) against the same calls on a FRESH engine each: sizes
from 50 to 250 000 reads in any order, aligned / unaligned / perfect, FASTA / FASTQ, -k 0 / 3 / 5, chimeric or not, with and without error
profile, single calls and steps (ns_generate_step).  A read is a function of (seed, index) and a call must not depend on what the calls
before it left in the context's buffers: records, error profile and per-read structs must be identical (checksums), and nothing may fault.
    python scripts/stress_sequences.py [calls=40] [rng seed=1]"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nanosim_amd import engine as E, model as M, synth  # noqa: E402

SEED = 20260926
n_calls = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
prefix = os.path.join(tempfile.mkdtemp(prefix="nsstress_"), "training")
synth.write_model(prefix, synth.SynthModelSpec(n_train=200_000, seed=SEED), write_pkl=False)
mdl = M.load_model(prefix, fastq=True, homopolymer=True, chimeric=True)
bases = synth.synth_sequence(synth.ECOLI_LEN, SEED, n_frac=0.0005, iupac_frac=0.0002, lower_frac=0.02, hp_boost=0.005)
ref = M.Reference(["ecoli-like"], bases, np.array([0, synth.ECOLI_LEN], dtype=np.uint64), np.array([1], dtype=np.uint8))


def fresh():
    e = E.Engine(0); e.set_reference(ref); e.load_model(mdl)
    return e


def digest(b, p):
    def cs(a):
        a = np.ascontiguousarray(a).view(np.uint8).ravel()
        n8 = len(a) // 8 * 8
        w = a[:n8].view(np.uint64)
        return (int(np.bitwise_xor.reduce(w)) if n8 else 0, int(w.sum(dtype=np.uint64)) if n8 else 0, int(a[n8:].sum()), len(a))
    r = b.reads()
    return (cs(b.records()), cs(b.errlog()) if p.emit_errlog else None, cs(r["seq_len"]), cs(r["n_pieces"]), int(b.info.n_reads))


def params(kind, n, first):
    kw = dict(seed=SEED + int(rng.integers(0, 5)), first_read=first, n_reads=n, max_len=ref.max_chrom)
    if kind == E.NS_KIND_ALIGNED:
        kw.update(chimeric=bool(rng.integers(0, 2)), fastq=bool(rng.integers(0, 2)), kmer_bias=int(rng.choice([0, 0, 3, 5])), emit_errlog=bool(rng.integers(0, 2)))
    elif kind == E.NS_KIND_UNALIGNED:
        kw.update(kind=kind, fastq=bool(rng.integers(0, 2)))
    else:
        kw.update(kind=kind, fastq=bool(rng.integers(0, 2)), kmer_bias=int(rng.choice([0, 5])))
    return kw


sizes = [50, 700, 4096, 16_384, 20_000, 60_000, 130_000, 250_000]
eng = fresh()
bad = 0
for i in range(n_calls):
    n = int(rng.choice(sizes)); first = int(rng.integers(0, 10**7))
    step = rng.random() < 0.3
    kind = E.NS_KIND_ALIGNED if step else int(rng.choice([E.NS_KIND_ALIGNED, E.NS_KIND_ALIGNED, E.NS_KIND_UNALIGNED, E.NS_KIND_PERFECT]))
    kw = params(kind, n, first)
    if kw.get("kmer_bias") and n > 130_000:
        n = kw["n_reads"] = 130_000                                   # (keeps the -k scratch images small)
    ref_eng = fresh()
    if step:
        kwu = params(E.NS_KIND_UNALIGNED, max(1, n // 19), first + 3)
        pa, pu = E.make_params(**kw), E.make_params(**kwu)
        ba, bu = eng.generate_step(pa, pu)
        got = (digest(ba, pa), digest(bu, pu))
        exp = (digest(ref_eng.generate(E.make_params(**kw)), pa),)
        ref2 = fresh()
        exp = exp + (digest(ref2.generate(E.make_params(**kwu)), pu),)
        ref2.close()
    else:
        p = E.make_params(**kw)
        got = (digest(eng.generate(p), p),)
        exp = (digest(ref_eng.generate(E.make_params(**kw)), p),)
    ref_eng.close()
    ok = got == exp
    bad += 0 if ok else 1
    print("%3d %-5s n %6d %-110s %s" % (i, "step" if step else "call", n, {k: v for k, v in kw.items() if k not in ("max_len", "n_reads")}, "identical" if ok else "DIFFERENT"), flush=True)
eng.close()
print("stress sequences:", "all identical" if not bad else "%d calls DIFFER" % bad)
sys.exit(1 if bad else 0)
