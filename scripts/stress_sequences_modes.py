#!/usr/bin/env python3
"""scripts/stress_sequences.py for the other two modes: random sequences of METAGENOME worker calls on one long-lived engine, and of
TRANSCRIPTOME worker calls (with and without intron retention) on another, each call against the same call on a fresh engine: sizes
from 100 to 60 000 reads in any order, FASTA / FASTQ, -k 0 / 4 / 5, chimeric (metagenome), uracil (transcriptome), error profile or not,
aligned and unaligned.  Records, error profile, per-read structs (and the species' base counts) must be identical; nothing may fault.
    python scripts/stress_sequences_modes.py [calls per mode = 40] [rng seed = 1]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
from nanosim_amd import engine as E, intron_retention as IR, metagenome as MG, model as M, transcriptome as T  # noqa: E402

n_calls = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
PREFIX = os.path.join(ROOT, "tests", "golden", "model_small", "training")
sizes = [100, 900, 4096, 16_384, 25_000, 60_000]


def cs(a):
    a = np.ascontiguousarray(a).view(np.uint8).ravel()
    n8 = len(a) // 8 * 8
    w = a[:n8].view(np.uint64)
    return (int(np.bitwise_xor.reduce(w)) if n8 else 0, int(w.sum(dtype=np.uint64)) if n8 else 0, int(a[n8:].sum()), len(a))


def digest(b, p):
    r = b.reads()
    return (cs(b.records()), cs(b.errlog()) if p.emit_errlog else None, cs(r["seq_len"]), cs(r["n_pieces"]), int(b.info.n_reads))


bad = 0
# ---- metagenome ----
META = os.path.join(ROOT, "tests", "golden", "meta")
mdl = M.load_model(PREFIX, chimeric=True, homopolymer=True, fastq=True)
ref = MG.read_metagenome(os.path.join(META, "genome_list.tsv"), os.path.join(META, "dna_type_list.tsv"))
_, samples = MG.read_abundance(os.path.join(META, "abundance.tsv"), ref.species)
abun = samples[0]
infl = {sp: MG.inflate_abun(abun, sp, mdl.abun_inflation) for sp in abun}


def meta_engine():
    e = E.Engine(0); e.set_metagenome(ref, abun, infl); e.load_model(mdl)
    return e


eng = meta_engine()
for i in range(n_calls):
    n = int(rng.choice(sizes))
    kind = int(rng.choice([E.NS_KIND_ALIGNED, E.NS_KIND_ALIGNED, E.NS_KIND_UNALIGNED]))
    kw = dict(seed=0xABC0 + int(rng.integers(0, 4)), first_read=int(rng.integers(0, 10**6)), n_reads=n, max_len=ref.max_chrom, meta=True, fastq=bool(rng.integers(0, 2)))
    if kind == E.NS_KIND_ALIGNED:
        kw.update(chimeric=bool(rng.integers(0, 2)), kmer_bias=int(rng.choice([0, 0, 4, 5])), emit_errlog=bool(rng.integers(0, 2)))
    else:
        kw.update(kind=kind)
    p = E.make_params(**kw)
    al = kind == E.NS_KIND_ALIGNED
    b = eng.generate(p); got = (digest(b, p), cs(eng.species_bases()) if al else None)
    f = meta_engine(); b2 = f.generate(E.make_params(**kw)); exp = (digest(b2, p), cs(f.species_bases()) if al else None); f.close()
    ok = got == exp; bad += 0 if ok else 1
    print("meta %3d n %6d %-120s %s" % (i, n, {k: v for k, v in kw.items() if k not in ("max_len", "n_reads", "meta")}, "identical" if ok else "DIFFERENT"), flush=True)
eng.close()

# ---- transcriptome ----
TRX = os.path.join(ROOT, "tests", "golden", "trx")
trx = T.read_transcriptome(os.path.join(TRX, "transcripts.fa"), os.path.join(TRX, "expression.tsv"), os.path.join(TRX, "polya.txt"), "guppy")
ir = IR.load(PREFIX, os.path.join(TRX, "genome.fa"), trx.ref)
tr_ir = T.restrict_expression(trx, ir.eligible)
mdl_t = M.load_model(PREFIX, transcriptome=True, fastq=True, homopolymer=True)


def trx_engine(use_ir):
    e = E.Engine(0); e.set_transcriptome(tr_ir if use_ir else trx)
    if use_ir:
        e.set_intron_retention(ir)
    e.load_model(mdl_t)
    return e


for use_ir in (False, True):
    eng = trx_engine(use_ir)
    for i in range(n_calls // 2):
        n = int(rng.choice(sizes))
        kind = int(rng.choice([E.NS_KIND_ALIGNED, E.NS_KIND_ALIGNED, E.NS_KIND_UNALIGNED]))
        kw = dict(seed=0x5EED + int(rng.integers(0, 4)), first_read=int(rng.integers(0, 10**6)), n_reads=n, max_len=10**9, trx=True, fastq=bool(rng.integers(0, 2)),
                  uracil=bool(rng.integers(0, 2)))
        if kind == E.NS_KIND_ALIGNED:
            kw.update(model_ir=use_ir, kmer_bias=int(rng.choice([0, 0, 4, 5])), emit_errlog=bool(rng.integers(0, 2)))
        else:
            kw.update(kind=kind)
        p = E.make_params(**kw)
        b = eng.generate(p); got = (digest(b, p), cs(b.polya()) if kind == E.NS_KIND_ALIGNED else None)
        f = trx_engine(use_ir); b2 = f.generate(E.make_params(**kw)); exp = (digest(b2, p), cs(b2.polya()) if kind == E.NS_KIND_ALIGNED else None); f.close()
        ok = got == exp; bad += 0 if ok else 1
        print("trx%s %3d n %6d %-120s %s" % (" IR" if use_ir else "   ", i, n, {k: v for k, v in kw.items() if k not in ("max_len", "n_reads", "trx")}, "identical" if ok else "DIFFERENT"), flush=True)
    eng.close()
print("stress sequences (metagenome, transcriptome):", "all identical" if not bad else "%d calls DIFFER" % bad)
sys.exit(1 if bad else 0)
