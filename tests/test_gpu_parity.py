"""GPU parity: the HIP path (through the C-ABI) must equal the CPU oracle bit-for-bit — same reads,
same per-read error CIGAR (events), same FASTA/FASTQ bytes, same error-profile rows."""
import numpy as np
import pytest

from nanosim_amd import engine as E
from nanosim_amd import model as M
from tests import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng(small_model, small_ref):
    e = E.Engine(0)
    e.set_reference(small_ref)
    e.load_model(small_model)
    yield e
    e.close()


def compare(batch, exp, params):
    reads, pieces, events = batch.reads(), batch.pieces(), batch.events()
    er, ep, ee = exp["reads"], exp["pieces"], exp["events"]
    assert len(reads) == len(er)
    fields = ("n_pieces", "reversed", "flags", "head", "tail", "seq_len", "attempts")
    for f in fields + (("rec_off",) if params.emit_records else ()):        # the oracle always formats records
        assert np.array_equal(reads[f], er[f]), f
    # the pieces of every read, through the read's own piece_off (a chimeric read that drew a new segment count in a later pass sits in
    # fresh slots behind the planned ones on the GPU; the oracle files its pieces one read after the other)
    def gather(rd, pc):
        k = rd["n_pieces"].astype(np.int64)
        idx = np.repeat(rd["piece_off"].astype(np.int64) - np.concatenate([[0], np.cumsum(k)[:-1]]), k) + np.arange(int(k.sum()))
        return pc[idx]
    gp, xp = gather(reads, pieces), gather(er, ep)
    assert len(gp) == len(xp) == len(ep)
    for f in ("ref_gpos", "chrom", "pos", "ref_len", "out_len", "n_ev", "kind"):
        assert np.array_equal(gp[f], xp[f]), f
    # per-piece CIGAR: identical event lists
    for i in range(len(gp)):
        g = events[int(gp["ev_off"][i]):int(gp["ev_off"][i]) + int(gp["n_ev"][i])]
        x = ee[int(xp["ev_off"][i]):int(xp["ev_off"][i]) + int(xp["n_ev"][i])]
        assert np.array_equal(g["pos"], x["pos"]) and np.array_equal(g["info"], x["info"]), "events of piece %d" % i
    assert int(batch.info.events_used) == len(ee)
    assert int(batch.info.total_bases) == exp["total_bases"]
    assert int(batch.info.total_ref_bases) == exp["total_ref_bases"]
    if not params.emit_records:
        assert batch.info.record_bytes == 0
        return
    rec = batch.records()
    assert rec.tobytes() == exp["records"].tobytes()
    if params.emit_errlog:
        assert batch.errlog().tobytes() == exp["errlog"].tobytes()


CASES = [
    dict(kind=E.NS_KIND_ALIGNED, n_reads=300, emit_errlog=True),
    dict(kind=E.NS_KIND_ALIGNED, n_reads=200, fastq=True, emit_errlog=True),
    dict(kind=E.NS_KIND_ALIGNED, n_reads=300, chimeric=True, fastq=True, emit_errlog=True),
    dict(kind=E.NS_KIND_UNALIGNED, n_reads=200, fastq=True),
    dict(kind=E.NS_KIND_UNALIGNED, n_reads=100),
    dict(kind=E.NS_KIND_PERFECT, n_reads=200, fastq=True),
    dict(kind=E.NS_KIND_ALIGNED, n_reads=150, min_len=3000, max_len=9000),            # rejections + epochs
    dict(kind=E.NS_KIND_ALIGNED, n_reads=150, median_len=5000, sd_len=0.4),            # -med/-sd
    dict(kind=E.NS_KIND_UNALIGNED, n_reads=100, median_len=800, sd_len=0.5),
    dict(kind=E.NS_KIND_UNALIGNED, n_reads=24, median_len=20000, sd_len=0.3, fastq=True),          # hundreds of 64-iteration blocks per read
    dict(kind=E.NS_KIND_UNALIGNED, n_reads=300, median_len=60, sd_len=0.6, min_len=1),              # reads shorter than one block
    dict(kind=E.NS_KIND_ALIGNED, n_reads=1, first_read=(1 << 33) + 5, emit_errlog=True),
    dict(kind=E.NS_KIND_ALIGNED, n_reads=0),
    dict(kind=E.NS_KIND_ALIGNED, n_reads=200, kmer_bias=5, emit_errlog=True),                      # -hp -k 5
    dict(kind=E.NS_KIND_ALIGNED, n_reads=200, kmer_bias=5, fastq=True, chimeric=True, emit_errlog=True),
    dict(kind=E.NS_KIND_ALIGNED, n_reads=4000, kmer_bias=4, fastq=True, min_len=2000, max_len=9000),   # final-length re-check + batch re-run
    dict(kind=E.NS_KIND_ALIGNED, n_reads=100, kmer_bias=6, emit_records=False),
    dict(kind=E.NS_KIND_PERFECT, n_reads=50, kmer_bias=5),                                          # -k is ignored with --perfect
    dict(kind=E.NS_KIND_ALIGNED, n_reads=100, kmer_bias=17, fastq=True),                            # k > 16: the thread-per-read homopolymer kernels
    dict(kind=E.NS_KIND_ALIGNED, n_reads=300, kmer_bias=3, fastq=True, emit_errlog=True),           # short k: runs everywhere
]


@pytest.mark.parametrize("case", CASES)
def test_gpu_equals_oracle(eng, small_model, small_ref, case):
    kw = dict(seed=0x5EED1234ABCD, first_read=0, max_len=small_ref.max_chrom)
    kw.update(case)
    p = E.make_params(**kw)
    b = eng.generate(p)
    if p.n_reads == 0:
        assert b.info.record_bytes == 0
        return
    big = p.use_lognormal and p.median_len >= 10000
    exp = O.generate(small_model, small_ref, p, bytes_per_read=400000 if big else 40000, events_per_read=60000 if big else 4000)
    compare(b, exp, p)


def test_gpu_circular_reference(small_model, circ_ref):
    e = E.Engine(0)
    try:
        e.set_reference(circ_ref)
        e.load_model(small_model)
        p = E.make_params(seed=77, first_read=0, n_reads=200, max_len=circ_ref.max_chrom, emit_errlog=True)
        b = e.generate(p)
        exp = O.generate(small_model, circ_ref, p)
        compare(b, exp, p)
        # wrap-around reads exist in this sample
        pc = b.pieces()
        assert np.any(pc["pos"].astype(np.int64) + pc["ref_len"] > circ_ref.genome_len)
        # -hp -k on reads across the origin (their tiles there take the generic kernel in both record passes)
        for kw in (dict(n_reads=300, kmer_bias=5, fastq=True, emit_errlog=True), dict(n_reads=300, kmer_bias=3), dict(n_reads=200, kmer_bias=4, chimeric=True, fastq=True)):
            p = E.make_params(seed=78, first_read=0, max_len=circ_ref.max_chrom, **kw)
            b = e.generate(p)
            compare(b, O.generate(small_model, circ_ref, p), p)
            pc = b.pieces()
            assert np.any(pc["pos"].astype(np.int64) + pc["ref_len"] > circ_ref.genome_len)
    finally:
        e.close()


def test_homopolymer_stage_at_every_k(small_model, small_ref):
    """mutate_homo's run scan and drain (S:627-705) at k = 2 .. 17 — the window logic of k_hp_scan changes with k (k < 4: runs straight to
    the run buffer; k > 16: only the run closed by a chunk's first start) — with chimeric reads (pieces that start inside a 16-byte group
    of the scratch image) and FASTQ class bits on the bases"""
    e = E.Engine(0)
    try:
        e.set_reference(small_ref)
        e.load_model(small_model)
        for k in (2, 3, 4, 5, 7, 9, 16, 17):
            for kw in (dict(n_reads=150, fastq=True, emit_errlog=True), dict(n_reads=100, chimeric=True)):
                p = E.make_params(seed=9000 + k, first_read=0, max_len=small_ref.max_chrom, kmer_bias=k, **kw)
                compare(e.generate(p), O.generate(small_model, small_ref, p), p)
    finally:
        e.close()


def test_gpu_circular_reference_unaligned_and_gaps(small_model, circ_ref):
    """unaligned reads and chimeric gaps (the dense record kernel) across the origin of a circular chromosome"""
    e = E.Engine(0)
    try:
        e.set_reference(circ_ref)
        e.load_model(small_model)
        for kw in (dict(kind=E.NS_KIND_UNALIGNED, n_reads=400, fastq=True), dict(kind=E.NS_KIND_UNALIGNED, n_reads=300, median_len=9000, sd_len=0.3),
                   dict(kind=E.NS_KIND_ALIGNED, n_reads=300, chimeric=True, emit_errlog=True)):
            p = E.make_params(seed=1234, first_read=0, max_len=circ_ref.max_chrom, **kw)
            b = e.generate(p)
            compare(b, O.generate(small_model, circ_ref, p, bytes_per_read=100000, events_per_read=20000), p)
            if kw["kind"] == E.NS_KIND_UNALIGNED:
                pc = b.pieces()
                assert np.any(pc["pos"].astype(np.int64) + pc["ref_len"] > circ_ref.genome_len)       # some reads wrap around
    finally:
        e.close()


def test_fractional_value_edges_take_the_global_tables():
    """The LDS image of the chain tables holds the ECDF value edges as 32-bit integers (whole numbers in every model read_analysis.py
    writes); a table with fractional edges (read_ecdf parses them as floats, S:69-97) keeps fp64 edges and the chain reads them from
    global memory — same reads as the oracle on the same table."""
    import copy
    import os
    from tests.conftest import GOLDEN
    mdl = M.load_model(os.path.join(GOLDEN, "model_small", "training"), chimeric=True, homopolymer=True, fastq=True)
    mdl = copy.deepcopy(mdl)
    mdl.first_match.vhi = np.asarray(mdl.first_match.vhi, dtype=np.float64) + 0.25
    for col in mdl.match_markov:
        col.vhi = np.asarray(col.vhi, dtype=np.float64) + 0.5
    ref = M.read_fasta(os.path.join(GOLDEN, "genome_small.fa"), "linear")
    e = E.Engine(0)
    try:
        e.set_reference(ref)
        e.load_model(mdl)
        for kw in (dict(n_reads=300, emit_errlog=True), dict(n_reads=200, chimeric=True, fastq=True)):
            p = E.make_params(seed=4242, first_read=0, max_len=ref.max_chrom, **kw)
            compare(e.generate(p), O.generate(mdl, ref, p), p)
    finally:
        e.close()


def test_reads_do_not_depend_on_batching(eng, small_ref):
    """(seed, read index) fully determine a read: a sub-range reproduces the same bytes."""
    p_all = E.make_params(seed=99, first_read=0, n_reads=300, max_len=small_ref.max_chrom)
    b = eng.generate(p_all)
    reads, rec = b.reads(), b.records()
    p_sub = E.make_params(seed=99, first_read=100, n_reads=100, max_len=small_ref.max_chrom)
    b2 = eng.generate(p_sub)
    reads2, rec2 = b2.reads(), b2.records()
    lo, hi = int(reads["rec_off"][100]), int(reads["rec_off"][200])
    assert rec2.tobytes() == rec[lo:hi].tobytes()
    assert np.array_equal(reads2["seq_len"], reads["seq_len"][100:200])


def test_error_paths(eng, small_ref):
    e2 = E.Engine(0)
    try:
        e2.set_reference(small_ref)
        import os
        from tests.conftest import GOLDEN
        e2.load_model(M.load_model(os.path.join(GOLDEN, "model_small", "training")))      # no -hp tables
        p = E.make_params(seed=1, first_read=0, n_reads=10, max_len=small_ref.max_chrom, kmer_bias=5)
        with pytest.raises(E.EngineError):
            e2.generate(p)
    finally:
        e2.close()
    p = E.make_params(seed=1, first_read=0, n_reads=10, min_len=10 ** 7, max_len=10 ** 8)
    with pytest.raises(E.EngineError):
        eng.generate(p)


def test_cooperative_chain_equals_oracle(small_model, small_ref, monkeypatch):
    """The wave-per-read chain used for the longest reads of big batches, forced onto half of a small batch."""
    monkeypatch.setenv("NS_COOP_MIN", "1")
    monkeypatch.setenv("NS_COOP_SHIFT", "1")
    e = E.Engine(0)
    try:
        e.set_reference(small_ref)
        e.load_model(small_model)
        for kw in (dict(n_reads=500, emit_errlog=True), dict(n_reads=400, chimeric=True, fastq=True, emit_errlog=True),
                   dict(n_reads=300, min_len=3000, max_len=9000), dict(n_reads=300, kmer_bias=5, fastq=True, emit_errlog=True)):
            args = dict(seed=31337, first_read=7, max_len=small_ref.max_chrom)
            args.update(kw)
            p = E.make_params(**args)
            compare(e.generate(p), O.generate(small_model, small_ref, p), p)
    finally:
        e.close()


def test_background_context_gives_the_same_unaligned_reads(small_model, small_ref, circ_ref, monkeypatch):
    """NS_UCOOP_SHIFT=3 (the split of a background context until round 6): all but the longest eighth of a batch of unaligned reads take
    the thread-per-read error list instead of the wave-per-read one — same reads, byte for byte"""
    monkeypatch.setenv("NS_COOP_MIN", "1")
    monkeypatch.setenv("NS_UCOOP_SHIFT", "3")
    for ref in (small_ref, circ_ref):
        e = E.Engine(0)
        try:
            e.set_background(True)
            e.set_reference(ref)
            e.load_model(small_model)
            for kw in (dict(n_reads=700, fastq=True), dict(n_reads=300, median_len=7000, sd_len=0.6), dict(n_reads=400, min_len=500, max_len=4000)):
                p = E.make_params(seed=2718, first_read=3, kind=E.NS_KIND_UNALIGNED, **{**dict(max_len=ref.max_chrom), **kw})
                compare(e.generate(p), O.generate(small_model, ref, p, bytes_per_read=200000, events_per_read=40000), p)
        finally:
            e.close()


@pytest.mark.parametrize("k, background", [(0, False), (0, True), (1, False)])
def test_wave_per_read_unaligned_chain_equals_oracle(small_model, small_ref, circ_ref, monkeypatch, k, background):
    """The wave-per-read unaligned chain with several loop iterations of S:1797-1829 per lane (NS_UCOOP_ITER of the build: 192 per round; round 6)
    and with one (NS_UCOOP_K=1,
    the form until then): reads of a few hundred bases end inside the first round (most lanes idle), 20-80 kb reads run hundreds of rounds;
    with NS_UCOOP_SHIFT=3 the longest eighth only, the rest thread per read."""
    monkeypatch.setenv("NS_COOP_MIN", "1")
    monkeypatch.setenv("NS_UCOOP_K", str(k))
    if background:
        monkeypatch.setenv("NS_UCOOP_SHIFT", "3")
    for ref in (small_ref, circ_ref):
        e = E.Engine(0)
        try:
            if background:
                e.set_background(True)
            e.set_reference(ref)
            e.load_model(small_model)
            for kw in (dict(n_reads=700, fastq=True), dict(n_reads=200, median_len=20000, sd_len=0.5), dict(n_reads=400, min_len=500, max_len=4000),
                       dict(n_reads=3, median_len=60000, sd_len=0.1), dict(n_reads=300, median_len=150, sd_len=0.8)):
                p = E.make_params(seed=161803, first_read=5, kind=E.NS_KIND_UNALIGNED, **{**dict(max_len=ref.max_chrom), **kw})
                compare(e.generate(p), O.generate(small_model, ref, p, bytes_per_read=400000, events_per_read=80000), p)
        finally:
            e.close()


def test_large_tables_of_a_trained_model_shape(tmp_path, small_ref, monkeypatch):
    """A model shaped like a real trained one (15 previous-match bins, 1500-row ECDFs: ~360 KB of chain tables).  Round 6: the LDS image
    holds the HOT PREFIX of every match-length column (the segments a draw reaches with probability >= 1 - 2^-12: ~43 KB, 512-thread
    workgroups), a draw behind a prefix takes the full column in global memory.  The same reads with prefixes cut at 1/8 of the probability
    (NS_TAIL_BITS=3: the full-column path under load), with 256-thread workgroups, and with the whole integer image read from global
    memory (an image forced out of LDS); the cooperative chain is forced on as well."""
    from nanosim_amd import synth
    bins = ((0, 1), (1, 2), (2, 3), (3, 5), (5, 7), (7, 10), (10, 14), (14, 19), (19, 25), (25, 33), (33, 45), (45, 60),
            (60, 90), (90, 150), (150, 1500))
    spec = synth.SynthModelSpec(n_train=3000, seed=99, ecdf_rows=1500, mm_bins=bins,
                                mm_means=tuple(20.0 + 2 * i for i in range(15)), mm_zero=(0.0,) + (0.02,) * 14, fm_mean=25.0)
    prefix = str(tmp_path / "big" / "training")
    synth.write_model(prefix, spec, write_pkl=False)
    mdl = M.load_model(prefix, chimeric=True, homopolymer=True, fastq=True)
    assert sum(len(c.hi) for c in mdl.match_markov) * 16 > 64 * 1024
    monkeypatch.setenv("NS_COOP_MIN", "1")
    monkeypatch.setenv("NS_COOP_SHIFT", "2")
    exp = {}
    for env in ({}, {"NS_TAIL_BITS": "3"}, {"NS_CHAIN_BLOCK": "256"}, {"NS_TAIL_BITS": "31"}):      # 31 bits: prefixes = (nearly) the full columns, no LDS
        for k in ("NS_TAIL_BITS", "NS_CHAIN_BLOCK"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        e = E.Engine(0)
        try:
            e.set_reference(small_ref)
            e.load_model(mdl)
            for i, kw in enumerate((dict(n_reads=600, emit_errlog=True), dict(n_reads=300, fastq=True, chimeric=True, kmer_bias=5))):
                args = dict(seed=4242, first_read=0, max_len=small_ref.max_chrom)
                args.update(kw)
                p = E.make_params(**args)
                if i not in exp:
                    exp[i] = O.generate(mdl, small_ref, p)
                compare(e.generate(p), exp[i], p)
        finally:
            e.close()


def test_dense_events_and_long_payloads(tmp_path, small_ref, circ_ref):
    """A model with an event every ~3 bases, runs of zero-length matches and long insertions: tiles that end early because more
    than 63 events start inside them, payloads that cross tile borders, letters beyond the first Philox word (> 16 per event),
    deletion-heavy spans, on a linear and on a circular reference (tiles beyond / across the origin)."""
    from nanosim_amd import synth
    spec = synth.SynthModelSpec(n_train=3000, seed=7, aligned_median=2500.0, mis=(3.0, 0.0, 0.3, 0.5), ins=(8.0, 0.9, 0.12, 0.5),
                                dele=(6.0, 0.95, 0.15, 0.5), mm_means=(2.0, 2.5, 3.0, 3.0, 3.5, 3.5, 4.0, 4.0),
                                mm_zero=(0.0, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3), fm_mean=3.0)
    prefix = str(tmp_path / "dense" / "training")
    synth.write_model(prefix, spec, write_pkl=False)
    mdl = M.load_model(prefix, chimeric=True, homopolymer=True, fastq=True)
    for ref in (small_ref, circ_ref):
        e = E.Engine(0)
        try:
            e.set_reference(ref)
            e.load_model(mdl)
            for kw in (dict(n_reads=300, emit_errlog=True), dict(n_reads=200, fastq=True, chimeric=True, emit_errlog=True),
                       dict(n_reads=150, kmer_bias=5, fastq=True), dict(n_reads=100, kind=E.NS_KIND_UNALIGNED, fastq=True)):
                args = dict(seed=987654321, first_read=3, max_len=ref.max_chrom)
                args.update(kw)
                p = E.make_params(**args)
                b = e.generate(p)
                exp = O.generate(mdl, ref, p)
                compare(b, exp, p)
                ev, pc = b.events(), b.pieces()
                if kw.get("kind", E.NS_KIND_ALIGNED) == E.NS_KIND_ALIGNED and not kw.get("kmer_bias"):
                    assert int(pc["n_ev"].sum()) > 0.15 * int(pc["ref_len"].sum())          # an event every few bases
                    assert int(M.ev_len(ev["info"]).max()) > 20                              # letters beyond the first word
        finally:
            e.close()


def test_later_attempt_outgrows_the_planned_event_capacity(tmp_path, small_ref):
    """Unaligned reads redraw their length at every attempt: a read whose attempt 0 is tiny (rejected by min_len) and whose next
    attempt is long needs more event slots than attempt 0 planned: every pass plans its own region of the event buffer."""
    from nanosim_amd import synth
    spec = synth.SynthModelSpec(n_train=4000, seed=3, unaligned_median=400.0, unaligned_sigma=1.6)     # wide: many draws below min_len
    prefix = str(tmp_path / "wide" / "training")
    synth.write_model(prefix, spec, write_pkl=False)
    mdl = M.load_model(prefix)
    e = E.Engine(0)
    try:
        e.set_reference(small_ref)
        e.load_model(mdl)
        p = E.make_params(seed=11, first_read=0, n_reads=3000, kind=E.NS_KIND_UNALIGNED, min_len=300, max_len=small_ref.max_chrom)
        b = e.generate(p)
        exp = O.generate(mdl, small_ref, p)
        compare(b, exp, p)
        rd = b.reads()
        assert int(rd["attempts"].max()) >= 2 and int(b.info.n_overflow) == 0          # no re-plan of the batch needed
    finally:
        e.close()


def test_event_capacity_overflow_replans_the_batch(small_model, small_ref, monkeypatch):
    """The recovery path of ns_generate: the chain finds more events than the batch planned slots for, the batch is planned again
    with twice the rate (nanosim_amd.hip: `cap_rate *= 2.0`).  NS_CAP_RATE_SCALE (a test knob read by ns_load_model) plans a
    twentieth of the model's rate, so the first plans MUST overflow; the batch that comes out equals the oracle's."""
    monkeypatch.setenv("NS_CAP_RATE_SCALE", "0.05")
    e = E.Engine(0)
    try:
        e.set_reference(small_ref)
        e.load_model(small_model)
        for kw in (dict(n_reads=400, emit_errlog=True), dict(n_reads=300, chimeric=True, fastq=True, emit_errlog=True),
                   dict(n_reads=300, kmer_bias=5, fastq=True, emit_errlog=True), dict(n_reads=300, min_len=3000, max_len=9000)):
            args = dict(seed=777, first_read=3, max_len=small_ref.max_chrom)
            args.update(kw)
            p = E.make_params(**args)
            b = e.generate(p)
            assert int(b.info.n_overflow) > 0, "the knob did not force a re-plan"
            compare(b, O.generate(small_model, small_ref, p), p)
    finally:
        e.close()


def test_homopolymer_edit_capacity_overflow_repeats_the_stage(small_model, small_ref, monkeypatch):
    """The recovery path of the -k stage: more homopolymer edits in a piece than its slots (hp_ev_slot) hold -> stats[7], the scan /
    drain pair runs again with twice the capacity.  NS_HP_CAP_SHIFT plans 2^-8 of the usual capacity and one slot of slack."""
    monkeypatch.setenv("NS_HP_CAP_SHIFT", "8")
    e = E.Engine(0)
    try:
        e.set_reference(small_ref)
        e.load_model(small_model)
        for kw in (dict(n_reads=300, kmer_bias=5, fastq=True, emit_errlog=True), dict(n_reads=200, kmer_bias=3, chimeric=True)):
            args = dict(seed=778, first_read=0, max_len=small_ref.max_chrom)
            args.update(kw)
            p = E.make_params(**args)
            compare(e.generate(p), O.generate(small_model, small_ref, p), p)
    finally:
        e.close()


def test_many_contig_reference(small_model):
    """A fragmented assembly (3 000 contigs of 0.5-40 kb, a few empty): the start-position walk of extract_read (S:1767-1780) is
    a bisection on the device; reads longer than most contigs are redrawn until they fit."""
    from nanosim_amd import synth
    rng = np.random.default_rng(5)
    lens = np.concatenate([rng.integers(500, 40000, 2996), [0, 0, 1, 90000]]).astype(np.int64)
    rng.shuffle(lens)
    seq = synth.synth_sequence(int(lens.sum()), 9, iupac_frac=0.0005)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    ref = M.Reference(["ctg%d" % i for i in range(len(lens))], seq, off, np.zeros(len(lens), dtype=np.uint8))
    e = E.Engine(0)
    try:
        e.set_reference(ref)
        e.load_model(small_model)
        for kw in (dict(n_reads=400, emit_errlog=True), dict(n_reads=300, chimeric=True, fastq=True), dict(n_reads=200, kind=E.NS_KIND_UNALIGNED),
                   dict(n_reads=200, kind=E.NS_KIND_PERFECT)):
            args = dict(seed=77, first_read=0, max_len=ref.max_chrom)
            args.update(kw)
            p = E.make_params(**args)
            compare(e.generate(p), O.generate(small_model, ref, p), p)
    finally:
        e.close()


def _long_insertion_model(small_model, mean):
    """the small model with insertion runs ~ Geometric(1 / mean) (table capped at 4095, include/nanosim_amd.h): reads whose net insertion
    balance leaves the 18-bit shift field of ns_event become likely"""
    import copy
    m = copy.deepcopy(small_model)
    cdf = 1.0 - (1.0 - 1.0 / mean) ** np.arange(1, 4096)
    cdf[-1] = 1.0
    m.mix_cdf[1] = [cdf.copy(), cdf.copy()]            # NS_INS
    return m


def test_reads_that_do_not_fit_the_event_record_are_redrawn(small_model, small_ref):
    """An event outside the fields of the 8-byte record (shift beyond +-131071) used to fail the whole batch with NS_EINVAL; now that ATTEMPT
    of the read is dropped and the read draws new lengths (like a failed final length check, S:1429-1430).  Same reads as the oracle,
    which applies the same rule; thread-per-read and wave-per-read error lists agree."""
    e = E.Engine(0)
    try:
        e.set_reference(small_ref)
        m = _long_insertion_model(small_model, 150)
        e.load_model(m)
        p = E.make_params(seed=7, first_read=0, n_reads=150, kind=E.NS_KIND_UNALIGNED, max_len=10**7)
        exp = O.generate(m, small_ref, p, bytes_per_read=600000, events_per_read=40000)
        assert int((exp["reads"]["attempts"] > 0).sum()) >= 3            # (no length limit in reach: every redraw is a range redraw)
        for background in (False, True):
            e.set_background(background)
            b = e.generate(p)
            compare(b, exp, p)
            assert 3 <= int(b.info.n_range_redraws) <= int(exp["reads"]["attempts"].sum())
        e.set_background(False)
        m = _long_insertion_model(small_model, 2500)
        e.load_model(m)
        p = E.make_params(seed=7, first_read=0, n_reads=200, max_len=small_ref.max_chrom, emit_errlog=True)
        b = e.generate(p)
        assert int(b.info.n_range_redraws) > 0
        compare(b, O.generate(m, small_ref, p, bytes_per_read=400000, events_per_read=8000), p)
    finally:
        e.close()


@pytest.mark.parametrize("fastq", [False, True])
def test_dense_homopolymer_edits_take_the_slow_tiles(small_model, fastq):
    """-k on a reference that is nothing but runs of five, with a homopolymer model that re-samples every run to nothing: hundreds of
    adjacent deletions at ONE output offset — more than the 63 events a tile of the record kernel stages.  Until round 2 the second
    record pass failed the batch with NS_EINVAL ("edits too dense"); now those tiles take a per-byte kernel (k_materialise_slow_hpf).
    Same bytes as the oracle."""
    import copy
    m = copy.deepcopy(small_model)
    for base in ("AT", "CG"):                       # mu = -50 whatever the run length, sigma 0.1: int(round(max(0, x))) == 0 (S:644-666)
        m.hp[base] = dict(m.hp[base], const=-50.0, alpha1=0.0, betas=[0.0] * len(m.hp[base]["betas"]), intercept=0.1, slope=0.0)
    seq = np.frombuffer((b"AAAAACCCCCGGGGGTTTTT" * 3000), dtype=np.uint8).copy()
    ref = M.Reference(["runs"], seq, np.array([0, len(seq)], dtype=np.uint64), np.array([0], dtype=np.uint8))
    e = E.Engine(0)
    try:
        e.set_reference(ref)
        e.load_model(m)
        p = E.make_params(seed=11, first_read=0, n_reads=120, fastq=fastq, kmer_bias=5, min_len=1, max_len=ref.max_chrom, emit_errlog=True)
        b = e.generate(p)
        exp = O.generate(m, ref, p, bytes_per_read=60000, events_per_read=8000)
        assert b.records().tobytes() == exp["records"].tobytes()
        assert b.errlog().tobytes() == exp["errlog"].tobytes()
        rd, er = b.reads(), exp["reads"]
        assert np.array_equal(rd["seq_len"], er["seq_len"]) and np.array_equal(rd["attempts"], er["attempts"])
        pc = b.pieces()[rd["piece_off"]]
        assert int(np.max(pc["ref_len"].astype(np.int64) - pc["out_len"])) > 2000        # thousands of bases deleted inside one piece
    finally:
        e.close()


def test_chimeric_fastq_with_reads_planned_again(small_model, circ_ref):
    """FASTQ, chimeric, many segments per read (segment mean 2.5): reads whose segment count changes with their epoch move to piece slots
    behind the planned ones, so piece_off is not in read order — the class words between the record kernel and k_qualities must not be
    placed by it (round 3: a read's words were overwritten by a re-planned read's; found by scripts/parity_sweep.py, 1 chunk in 120)."""
    import copy
    m = copy.deepcopy(small_model)
    m.segment_mean = 2.5
    m.nseg_cdf = M.geometric_cdf(1.0 / m.segment_mean, 0)
    e = E.Engine(0)
    try:
        e.set_reference(circ_ref)
        e.load_model(m)
        moved = 0
        # (-k: the slots of a piece's homopolymer edits follow the order of the scratch buffer = read order; placed by piece_off, the
        # slots of a re-planned read overlapped another read's: garbage bases, a memory fault with k = 4)
        for first, kw in ((0, {}), (3000, {}), (0, dict(kmer_bias=4, emit_errlog=True)), (3000, dict(kmer_bias=5))):
            p = E.make_params(seed=0xC0FFEE, first_read=first, n_reads=3000, fastq=True, chimeric=True, max_len=circ_ref.max_chrom, **kw)
            b = e.generate(p)
            compare(b, O.generate(m, circ_ref, p, bytes_per_read=160000, events_per_read=24000), p)
            rd = b.reads()
            po = rd["piece_off"].astype(np.int64)
            moved += int(np.sum(po[1:] < po[:-1]))
        assert moved > 0                       # some reads lie out of order in the piece array
    finally:
        e.close()


def test_ecdf_segments_of_every_width(small_ref):
    """The LDS image of the chain answers an ECDF look-up without floating point where it can (ns_chain.h: ecdf_lookup_u): unit-wide
    segments by a flag, segments up to 15 units wide by a threshold list the host derives from the fp64 formula, wider ones (empty
    histogram bins merged, S:216-221) by the formula itself.  A model whose match-length columns mix all three, with edges that sit on
    and next to the 2^-32 grid of the draws: same reads as the oracle (which only knows the fp64 formula)."""
    import copy
    import os
    from tests.conftest import GOLDEN
    mdl = copy.deepcopy(M.load_model(os.path.join(GOLDEN, "model_small", "training"), chimeric=True, homopolymer=True, fastq=True))
    rng = np.random.default_rng(5)
    widths = [1, 1, 2, 1, 3, 1, 1, 7, 1, 15, 1, 16, 1, 1, 40, 2, 1, 1, 9, 1]
    for ci, col in enumerate(mdl.match_markov):
        w = np.array(widths[ci % 5:] + widths[:ci % 5], dtype=np.float64)
        mass = rng.random(len(w)) + 0.05
        hi = np.cumsum(mass / mass.sum())
        # some edges exactly on a draw's probability (u + 0.5) 2^-32 and one ulp beside it: where fp64 rounding decides the last step
        for k in range(0, len(hi) - 1, 3):
            u = np.floor(hi[k] * 4294967296.0)
            hi[k] = (u + 0.5) * 2.0 ** -32 if k % 2 == 0 else np.nextafter((u + 0.5) * 2.0 ** -32, 1.0)
        hi[-1] = 1.0
        col.hi = hi
        col.vhi = float(col.vlo0) + np.cumsum(w)
    e = E.Engine(0)
    try:
        e.set_reference(small_ref)
        e.load_model(mdl)
        for kw in (dict(n_reads=1500, emit_errlog=True), dict(n_reads=500, chimeric=True, fastq=True)):
            p = E.make_params(seed=99, first_read=0, max_len=small_ref.max_chrom, **kw)
            compare(e.generate(p), O.generate(mdl, small_ref, p, bytes_per_read=200000, events_per_read=40000), p)
    finally:
        e.close()
