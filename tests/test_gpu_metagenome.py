"""GPU parity for metagenome batches: the HIP path through the C-ABI equals the CPU oracle
bit for bit — pass structure, species assignment, read numbering, names with gap components, records, error rows and
the per-species base counts."""
import os

import numpy as np
import pytest

from nanosim_amd import engine as E
from nanosim_amd import metagenome as MG
from nanosim_amd import model as M
from tests import oracle_lib as O
from tests.test_gpu_parity import compare

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
META = os.path.join(ROOT, "tests", "golden", "meta")


@pytest.fixture(scope="module")
def meta_ref():
    cwd = os.getcwd()
    os.chdir(ROOT)
    try:
        return MG.read_metagenome(os.path.join(META, "genome_list.tsv"), os.path.join(META, "dna_type_list.tsv"))
    finally:
        os.chdir(cwd)


@pytest.fixture(scope="module")
def setup(small_model, meta_ref):
    _, samples = MG.read_abundance(os.path.join(META, "abundance.tsv"), meta_ref.species)
    abun = samples[0]
    infl = {sp: MG.inflate_abun(abun, sp, small_model.abun_inflation) for sp in abun}
    e = E.Engine(0)
    e.set_metagenome(meta_ref, abun, infl)
    e.load_model(small_model)
    yield e, abun, infl
    e.close()


CASES = [
    dict(n_reads=400, emit_errlog=True),
    dict(n_reads=300, fastq=True, emit_errlog=True),
    dict(n_reads=500, chimeric=True, fastq=True, emit_errlog=True),
    dict(n_reads=600, chimeric=True),
    dict(n_reads=300, min_len=3000, max_len=9000),                     # many passes: most reads are rejected per pass
    dict(n_reads=300, chimeric=True, min_len=2000, max_len=20000, emit_errlog=True),
    dict(n_reads=1, first_read=(1 << 33) + 11),
    dict(n_reads=250, emit_records=False),
    dict(n_reads=200, kind=E.NS_KIND_UNALIGNED, fastq=True),
    dict(n_reads=150, kind=E.NS_KIND_UNALIGNED, median_len=900, sd_len=0.4),
    dict(n_reads=300, kind=E.NS_KIND_PERFECT),                             # --perfect worker: no errors, quotas never updated
    dict(n_reads=200, kind=E.NS_KIND_PERFECT, fastq=True, min_len=4000, max_len=12000),
    dict(n_reads=300, kmer_bias=5, fastq=True, emit_errlog=True),                                  # -hp -k 5
    dict(n_reads=300, kmer_bias=5, chimeric=True, emit_errlog=True),
    dict(n_reads=2000, kmer_bias=4, min_len=2500, max_len=9000),                                   # final-length check after the homopolymer stage
    dict(n_reads=400, median_len=4000, sd_len=0.5, emit_errlog=True),                              # -med/-sd
    dict(n_reads=300, median_len=6000, sd_len=0.3, kind=E.NS_KIND_PERFECT),
]


@pytest.mark.parametrize("case", CASES)
def test_gpu_metagenome_equals_oracle(setup, small_model, meta_ref, case):
    eng, abun, infl = setup
    kw = dict(seed=0xFEED5EED77, first_read=0, max_len=meta_ref.max_chrom, meta=True)
    kw.update(case)
    p = E.make_params(**kw)
    mdl = small_model
    if p.kind == E.NS_KIND_PERFECT:          # the KDE of whole aligned reads replaces the aligned-region one (S:473-476)
        mdl = M.load_model(os.path.join(ROOT, "tests", "golden", "model_small", "training"), perfect=True, fastq=True)
        eng.load_model(mdl)
    try:
        b = eng.generate(p)
        exp = O.generate_meta(mdl, meta_ref, abun, infl if p.chimeric else None, p)
        compare(b, exp, p)
        if p.kind != E.NS_KIND_UNALIGNED:
            assert np.array_equal(eng.species_bases(), exp["species_bases"])
    finally:
        if mdl is not small_model:
            eng.load_model(small_model)


def test_metagenome_event_capacity_overflow_replans_the_pass(small_model, meta_ref, monkeypatch):
    """The metagenome worker's twin of the event-capacity re-plan (meta_passes: `P.cap_rate *= 2.0`), forced by NS_CAP_RATE_SCALE
    (a test knob read by ns_load_model: it plans a twentieth of the model's event rate)."""
    monkeypatch.setenv("NS_CAP_RATE_SCALE", "0.05")
    _, samples = MG.read_abundance(os.path.join(META, "abundance.tsv"), meta_ref.species)
    abun = samples[0]
    infl = {sp: MG.inflate_abun(abun, sp, small_model.abun_inflation) for sp in abun}
    e = E.Engine(0)
    try:
        e.set_metagenome(meta_ref, abun, infl)
        e.load_model(small_model)
        for kw in (dict(n_reads=400, emit_errlog=True), dict(n_reads=300, chimeric=True, fastq=True)):
            p = E.make_params(seed=0xFEED5EED78, first_read=0, max_len=meta_ref.max_chrom, meta=True, **kw)
            b = e.generate(p)
            assert int(b.info.n_overflow) > 0, "the knob did not force a re-plan"
            compare(b, O.generate_meta(small_model, meta_ref, abun, infl if p.chimeric else None, p), p)
    finally:
        e.close()


def test_metagenome_batches_are_reproducible(setup, meta_ref):
    eng, abun, infl = setup
    p = E.make_params(seed=5, first_read=1000, n_reads=2000, chimeric=True, fastq=True, max_len=meta_ref.max_chrom, meta=True)
    a = eng.generate(p).records().tobytes()
    assert eng.generate(p).records().tobytes() == a
    # a genome-mode batch on the same engine afterwards is unaffected by the metagenome state
    g = E.make_params(seed=5, first_read=0, n_reads=50, max_len=20000)
    assert eng.generate(g).records().tobytes() == eng.generate(g).records().tobytes()


def test_metagenome_error_paths(small_model, small_ref, meta_ref, setup):
    eng, abun, infl = setup
    with pytest.raises(E.EngineError):
        eng.generate(E.make_params(seed=1, first_read=0, n_reads=10, max_len=9000, meta=True, kind=E.NS_KIND_PERFECT, chimeric=True))
    e2 = E.Engine(0)
    try:
        e2.set_reference(small_ref)
        e2.load_model(small_model)
        with pytest.raises(E.EngineError):                       # no species view
            e2.generate(E.make_params(seed=1, first_read=0, n_reads=10, max_len=9000, meta=True))
    finally:
        e2.close()


def test_genome_list_may_hold_species_the_abundance_table_does_not_name(small_model, meta_ref):
    """The reference's quotas run over dict_abun only (S:772-775): a genome of the list without an abundance row gets no aligned
    reads (its chromosomes still serve gaps and unaligned reads)."""
    _, samples = MG.read_abundance(os.path.join(META, "abundance.tsv"), meta_ref.species)
    abun = dict(samples[0])
    dropped = sorted(abun)[0]
    del abun[dropped]
    e = E.Engine(0)
    try:
        e.set_metagenome(meta_ref, abun, None)
        e.load_model(small_model)
        p = E.make_params(seed=5, first_read=0, n_reads=500, meta=True, max_len=meta_ref.max_chrom)
        b = e.generate(p)
        bases = e.species_bases()
        assert bases[meta_ref.species.index(dropped)] == 0 and bases.sum() > 0
        names = [ln for ln in b.records().tobytes().split(b"\n") if ln.startswith(b">")]
        assert len(names) == 500 and not any(nm[1:].startswith(dropped.encode() + b"-") for nm in names)
    finally:
        e.close()


def test_many_pass_chimeric_batch_keeps_its_pieces_in_bounds(small_model, setup):
    """A narrow length window rejects most chimeric reads of a pass; later passes re-sort the remaining segment counts, so the pieces
    accepted over all passes can outnumber the first plan: the buffer grows, GPU == oracle."""
    e, abun, infl = setup
    p = E.make_params(seed=4242, first_read=0, n_reads=3000, chimeric=True, meta=True, min_len=1500, max_len=6000)
    b = e.generate(p)
    reads, pieces = b.reads(), b.pieces()
    assert int(reads["n_pieces"].sum()) == len(pieces) == int(b.info.n_pieces)
    assert np.all(reads["piece_off"].astype(np.int64) + reads["n_pieces"] <= len(pieces))
