"""GPU parity for transcriptome batches: expression-weighted transcript pick, aligned length from the 2-D KDE
sample kept until a transcript repeats (S:1080-1104, per block of 1 024 read indices), start inside the transcript, polyA tails,
--uracil, --perfect, unaligned reads — bit for bit against the CPU oracle."""
import os

import numpy as np
import pytest

from nanosim_amd import engine as E
from nanosim_amd import intron_retention as IR
from nanosim_amd import model as M
from nanosim_amd import transcriptome as T
from tests import oracle_lib as O
from tests.test_gpu_parity import compare

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TRX = os.path.join(ROOT, "tests", "golden", "trx")
PREFIX = os.path.join(ROOT, "tests", "golden", "model_small", "training")


@pytest.fixture(scope="module")
def trx_ref():
    return T.read_transcriptome(os.path.join(TRX, "transcripts.fa"), os.path.join(TRX, "expression.tsv"), os.path.join(TRX, "polya.txt"), "guppy")


CASES = [
    dict(n_reads=500, emit_errlog=True),
    dict(n_reads=400, fastq=True, emit_errlog=True),
    dict(n_reads=300, fastq=True, uracil=True),
    dict(n_reads=300, kind=E.NS_KIND_PERFECT, fastq=True),
    dict(n_reads=300, kind=E.NS_KIND_PERFECT, uracil=True),
    dict(n_reads=300, kind=E.NS_KIND_UNALIGNED, fastq=True, min_len=50, max_len=5000),
    dict(n_reads=200, kind=E.NS_KIND_UNALIGNED, uracil=True, min_len=50, max_len=5000),
    dict(n_reads=1, first_read=(1 << 34) + 3),
    dict(n_reads=300, emit_records=False),
    dict(n_reads=300, kmer_bias=5, fastq=True, uracil=True, emit_errlog=True),        # -hp -k 5
    dict(n_reads=300, kmer_bias=4),
    dict(n_reads=2500, fastq=True, emit_errlog=True),                                  # three blocks of the sample-until-repeat walk
    dict(n_reads=1500, first_read=700, emit_errlog=True),                              # starts and ends inside a block (dry runs in front)
    dict(n_reads=1100, first_read=5 * 1024, kind=E.NS_KIND_PERFECT),
]


@pytest.mark.parametrize("case", CASES)
def test_gpu_transcriptome_equals_oracle(trx_ref, case):
    kw = dict(seed=0xABCD1234, first_read=0, max_len=10 ** 9, trx=True)
    kw.update(case)
    p = E.make_params(**kw)
    mdl = M.load_model(PREFIX, transcriptome=True, perfect=p.kind == E.NS_KIND_PERFECT, fastq=True, homopolymer=p.kind != E.NS_KIND_PERFECT)
    e = E.Engine(0)
    try:
        e.set_transcriptome(trx_ref)
        e.load_model(mdl)
        b = e.generate(p)
        exp = O.generate_trx(mdl, trx_ref, p)
        compare(b, exp, p)
        assert np.array_equal(b.polya(), exp["polya"])
        if p.kind != E.NS_KIND_UNALIGNED:
            assert int(b.polya().max()) >= 2 or p.n_reads < 10
    finally:
        e.close()


def test_transcriptome_event_capacity_overflow_replans_the_batch(trx_ref, monkeypatch):
    """The transcriptome worker's twin of the event-capacity re-plan, forced by NS_CAP_RATE_SCALE (test knob, ns_load_model)."""
    monkeypatch.setenv("NS_CAP_RATE_SCALE", "0.05")
    mdl = M.load_model(PREFIX, transcriptome=True, fastq=True, homopolymer=True)
    e = E.Engine(0)
    try:
        e.set_transcriptome(trx_ref)
        e.load_model(mdl)
        for kw in (dict(n_reads=500, emit_errlog=True), dict(n_reads=300, kmer_bias=5, fastq=True)):
            p = E.make_params(seed=0xABCD1235, first_read=0, max_len=10 ** 9, trx=True, **kw)
            b = e.generate(p)
            assert int(b.info.n_overflow) > 0, "the knob did not force a re-plan"
            compare(b, O.generate_trx(mdl, trx_ref, p), p)
    finally:
        e.close()


IR_CASES = [
    dict(n_reads=600, emit_errlog=True),
    dict(n_reads=500, fastq=True, uracil=True, emit_errlog=True),
    dict(n_reads=400, kmer_bias=5, fastq=True, emit_errlog=True),                       # -hp -k 5 on spliced stretches
    dict(n_reads=300, kmer_bias=4),
    dict(n_reads=300, kind=E.NS_KIND_PERFECT),                                           # S:1117: --perfect never retains introns
    dict(n_reads=300, kind=E.NS_KIND_UNALIGNED, min_len=50, max_len=5000),
    dict(n_reads=3000, emit_records=False),
    dict(n_reads=1300, first_read=1000, fastq=True, emit_errlog=True),                  # a batch that starts inside a block
]


@pytest.mark.parametrize("case", IR_CASES)
def test_gpu_intron_retention_equals_oracle(trx_ref, case):
    """S:114-191, 1156-1192: structure chain, genomic intervals, splice from the genome (both strands, soft-masked introns),
    read names with the retained stretches, polyA rule of extract_read_pos"""
    ir = IR.load(PREFIX, os.path.join(TRX, "genome.fa"), trx_ref.ref)
    tr = T.restrict_expression(trx_ref, ir.eligible)
    kw = dict(seed=0x77AB12, first_read=0, max_len=10 ** 9, trx=True, model_ir=True)
    kw.update(case)
    p = E.make_params(**kw)
    mdl = M.load_model(PREFIX, transcriptome=True, perfect=p.kind == E.NS_KIND_PERFECT, fastq=True, homopolymer=p.kind != E.NS_KIND_PERFECT)
    e = E.Engine(0)
    try:
        e.set_transcriptome(tr)
        e.set_intron_retention(ir)
        e.load_model(mdl)
        b = e.generate(p)
        exp = O.generate_trx(mdl, tr, p, ir=ir)
        compare(b, exp, p)
        assert np.array_equal(b.polya(), exp["polya"])
        n_spliced = int((b.pieces()["ref_gpos"] >= E.NS_SPLICED_BASE).sum())
        if p.kind == E.NS_KIND_ALIGNED:
            assert n_spliced > p.n_reads // 10
            got = b.spliced()
            assert len(got) == len(exp["spliced"])
            pcs = b.pieces()
            for pc in pcs[pcs["ref_gpos"] >= E.NS_SPLICED_BASE]:                          # (the padding between the stretches is not defined)
                o = int(pc["ref_gpos"] - E.NS_SPLICED_BASE)
                assert np.array_equal(got[o:o + int(pc["ref_len"])] & 0x7F, exp["spliced"][o:o + int(pc["ref_len"])])   # (bit 7: the engine's IUPAC mark)
        else:
            assert n_spliced == 0
        # the engine keeps working without intron retention afterwards
        p.model_ir = 0
        compare(e.generate(p), O.generate_trx(mdl, tr, p), p)
    finally:
        e.close()


def test_intron_retention_error_paths(trx_ref):
    e = E.Engine(0)
    try:
        e.set_transcriptome(trx_ref)
        e.load_model(M.load_model(PREFIX, transcriptome=True))
        with pytest.raises(E.EngineError):                       # no structure tables
            e.generate(E.make_params(seed=1, first_read=0, n_reads=10, max_len=10 ** 9, trx=True, model_ir=True))
        ir = IR.load(PREFIX, os.path.join(TRX, "genome.fa"), trx_ref.ref)
        e.set_intron_retention(ir)
        e.generate(E.make_params(seed=1, first_read=0, n_reads=10, max_len=10 ** 9, trx=True, model_ir=True))
        e.set_transcriptome(trx_ref)                             # a new reference drops the tables
        with pytest.raises(E.EngineError):
            e.generate(E.make_params(seed=1, first_read=0, n_reads=10, max_len=10 ** 9, trx=True, model_ir=True))
        with pytest.raises(E.EngineError):                       # not a transcriptome batch
            e.generate(E.make_params(seed=1, first_read=0, n_reads=10, max_len=9000, model_ir=True))
    finally:
        e.close()


def test_transcriptome_error_paths(trx_ref, small_model, small_ref):
    e = E.Engine(0)
    try:
        e.set_reference(small_ref)
        e.load_model(small_model)
        with pytest.raises(E.EngineError):                       # no expression view
            e.generate(E.make_params(seed=1, first_read=0, n_reads=10, max_len=9000, trx=True))
        e.set_transcriptome(trx_ref)
        with pytest.raises(E.EngineError):                       # the genome-mode model has no 2-D KDE
            e.generate(E.make_params(seed=1, first_read=0, n_reads=10, max_len=9000, trx=True))
        e.load_model(M.load_model(PREFIX, transcriptome=True))
        with pytest.raises(E.EngineError):                       # -k without the homopolymer model
            e.generate(E.make_params(seed=1, first_read=0, n_reads=10, max_len=9000, trx=True, kmer_bias=5))
        with pytest.raises(E.EngineError):
            e.generate(E.make_params(seed=1, first_read=0, n_reads=10, max_len=9000, trx=True, chimeric=True))
    finally:
        e.close()


def test_gpu_transcriptome_batches_do_not_depend_on_the_split(trx_ref):
    """the sample-until-repeat rule couples the reads of a BLOCK of read indices, not of a batch: any split of a run gives its bytes"""
    mdl = M.load_model(PREFIX, transcriptome=True, fastq=True)
    kw = dict(seed=0xFEED, max_len=10 ** 9, trx=True, fastq=True, emit_errlog=True)
    e = E.Engine(0)
    try:
        e.set_transcriptome(trx_ref)
        e.load_model(mdl)
        b = e.generate(E.make_params(first_read=0, n_reads=3100, **kw))
        whole, whole_err = b.records().tobytes(), b.errlog().tobytes()
        recs, errs = [], []
        for lo, hi in ((0, 1), (1, 1023), (1023, 1024), (1024, 2500), (2500, 3100)):
            b = e.generate(E.make_params(first_read=lo, n_reads=hi - lo, **kw))
            recs.append(b.records().tobytes()); errs.append(b.errlog().tobytes())
        assert b"".join(recs) == whole and b"".join(errs) == whole_err
    finally:
        e.close()
