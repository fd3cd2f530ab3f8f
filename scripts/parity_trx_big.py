"""Transcriptome worker calls far larger than the -m gpu cases (tests/test_gpu_transcriptome.py), with and without intron retention,
GPU == oracle bit for bit:  python scripts/parity_trx_big.py [reads per case = 30000]      (GPU box)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
from nanosim_amd import engine as E, intron_retention as IR, model as M, transcriptome as T  # noqa: E402
from tests import oracle_lib as O  # noqa: E402
from tests.test_gpu_parity import compare  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
TRX = os.path.join(ROOT, "tests", "golden", "trx")
PREFIX = os.path.join(ROOT, "tests", "golden", "model_small", "training")
trx = T.read_transcriptome(os.path.join(TRX, "transcripts.fa"), os.path.join(TRX, "expression.tsv"), os.path.join(TRX, "polya.txt"), "guppy")
ir = IR.load(PREFIX, os.path.join(TRX, "genome.fa"), trx.ref)
tr_ir = T.restrict_expression(trx, ir.eligible)
mdl = M.load_model(PREFIX, transcriptome=True, fastq=True, homopolymer=True)
bad = 0
for name, use_ir, kw in (("fastq uracil errlog", False, dict(fastq=True, uracil=True, emit_errlog=True)),
                         ("fastq -k5 errlog", False, dict(kmer_bias=5, fastq=True, emit_errlog=True)),
                         ("IR fastq uracil errlog", True, dict(fastq=True, uracil=True, emit_errlog=True)),
                         ("IR fastq -k5", True, dict(kmer_bias=5, fastq=True)),
                         ("IR fasta -k4 errlog", True, dict(kmer_bias=4, emit_errlog=True))):
    e = E.Engine(0)
    try:
        e.set_transcriptome(tr_ir if use_ir else trx)
        if use_ir:
            e.set_intron_retention(ir)
        e.load_model(mdl)
        p = E.make_params(seed=0x5EED1234, first_read=0, n_reads=n, max_len=10 ** 9, trx=True, model_ir=use_ir, **kw)
        b = e.generate(p)
        exp = O.generate_trx(mdl, tr_ir if use_ir else trx, p, ir=ir if use_ir else None)
        try:
            compare(b, exp, p)
            assert np.array_equal(b.polya(), exp["polya"])
            print("%-28s %6d reads  identical" % (name, n))
        except AssertionError as ex:
            bad += 1
            print("%-28s %6d reads  DIFFERS: %s" % (name, n, str(ex)[:200]))
    finally:
        e.close()
sys.exit(1 if bad else 0)
