"""Full-size checks on the references BASELINE.json's configs name (content is seeded random sequence):
configs[2] chr1_like (248 956 422 bp linear) with --fastq -hp -k 5, configs[3] grch38_like (24 chromosomes, 3.1 Gb: offsets beyond
2^32) with --chimeric, configs[4] zymo10_like (10 species, 45 chromosomes, the abundance columns of the reference's
sample_config_file/).  Size-independent properties (determinism, independence from batching, structural invariants of every read)
and byte-for-byte oracle checks of sampled read indices / of a whole metagenome worker batch."""
import numpy as np
import pytest

from nanosim_amd import engine as E
from nanosim_amd import metagenome as MG
from nanosim_amd import model as M
from nanosim_amd import synth
from tests import oracle_lib as O
from tests.test_gpu_fullsize import _record_slice, checksum

pytestmark = pytest.mark.gpu
SEED = 20260926


@pytest.fixture(scope="module")
def hg002_model(tmp_path_factory):
    prefix = str(tmp_path_factory.mktemp("hg002_like") / "training")
    synth.write_model(prefix, synth.SynthModelSpec(n_train=1_000_000, seed=SEED), write_pkl=False)
    return M.load_model(prefix, fastq=True, homopolymer=True, chimeric=True)


def _fastq_structure(rec, reads):
    """every record of a FASTQ image: '@name\\nSEQ\\n+\\nQUAL\\n' with len(SEQ) == len(QUAL) == seq_len, bases ACGT, qualities 1..93"""
    off = reads["rec_off"].astype(np.int64)
    L = reads["seq_len"].astype(np.int64)
    end = np.concatenate([off[1:], [len(rec)]])
    nl = end - off - 2 * L - 6                                   # name length
    assert np.all(nl > 10)
    assert np.all(rec[off] == ord("@")) and np.all(rec[off + nl + 1] == 10) and np.all(rec[off + nl + 2 + L] == 10)
    assert np.all(rec[off + nl + 3 + L] == ord("+")) and np.all(rec[off + nl + 4 + L] == 10) and np.all(rec[end - 1] == 10)
    hist = np.bincount(rec, minlength=256)
    newlines = 4 * len(reads)
    assert hist[10] == newlines
    return nl


def test_chr1_like_fastq_homopolymer_2m_reads(hg002_model):
    """configs[2]: 2 x 10^6 reads in two launches on the 249 Mb linear chromosome"""
    mdl = hg002_model
    seq = synth.synth_sequence(synth.CHR1_LEN, SEED, n_frac=0.0005, iupac_frac=0.0002, lower_frac=0.02, hp_boost=0.005)
    ref = M.Reference(["chr1-like"], seq, np.array([0, len(seq)], dtype=np.uint64), np.array([0], dtype=np.uint8))
    eng = E.Engine(0)
    try:
        eng.set_reference(ref)
        eng.load_model(mdl)
        n = 1_000_000
        kw = dict(seed=SEED + 3, fastq=True, kmer_bias=5, max_len=ref.max_chrom)
        sums = []
        for part in range(2):
            b = eng.generate(E.make_params(first_read=part * n, n_reads=n, **kw))
            reads, pieces = b.reads(), b.pieces()
            assert np.all(reads["flags"] == 0) and np.all(reads["n_pieces"] == 1)
            # linear chromosome (S:1767-1780): the segment lies inside it
            assert np.all(pieces["pos"].astype(np.int64) + pieces["ref_len"] <= synth.CHR1_LEN) and np.all(pieces["chrom"] == 0)
            assert np.array_equal(reads["seq_len"], reads["head"] + pieces["out_len"] + reads["tail"])
            assert int(b.info.total_bases) == int(reads["seq_len"].astype(np.int64).sum())
            assert pieces["pos"].max() > 200_000_000 and pieces["pos"].min() < 50_000_000          # the whole chromosome is used
            rec = b.records()
            _fastq_structure(rec, reads)
            sums.append(checksum(rec))
            rng = np.random.default_rng(11 + part)
            for r in sorted(set(int(x) for x in rng.integers(0, n, 10)) | {0, n - 1}):
                exp = O.generate(mdl, ref, E.make_params(first_read=part * n + r, n_reads=1, **kw), bytes_per_read=1_000_000, events_per_read=100_000)
                assert _record_slice(eng, b, reads, r, n) == exp["records"].tobytes(), (part, r)
            if part == 1:       # independence from batching: the second million again as two halves
                h = n // 2
                ra = eng.generate(E.make_params(first_read=n, n_reads=h, **kw)).records()
                assert checksum(ra) == checksum(rec[:int(reads["rec_off"][h])])
        assert sums[0] != sums[1]
        # the unaligned reads of the same run (S:1482-1549: the dense record kernel), 10^5 of them behind the aligned ones
        nu = 100_000
        ku = dict(seed=SEED + 3, fastq=True, kind=E.NS_KIND_UNALIGNED, max_len=ref.max_chrom)
        b = eng.generate(E.make_params(first_read=2 * n, n_reads=nu, **ku))
        reads, pieces = b.reads(), b.pieces()
        assert np.all(reads["flags"] == 0) and np.all(reads["n_pieces"] == 1) and np.all(pieces["kind"] == 1)
        assert np.all(pieces["pos"].astype(np.int64) + pieces["ref_len"] <= synth.CHR1_LEN)
        assert np.array_equal(reads["seq_len"], pieces["out_len"]) and int(b.info.total_bases) == int(reads["seq_len"].astype(np.int64).sum())
        rec = b.records()
        _fastq_structure(rec, reads)
        longest = int(np.argmax(reads["seq_len"]))
        for r in sorted({0, nu - 1, longest} | set(int(x) for x in np.random.default_rng(5).integers(0, nu, 6))):
            exp = O.generate(mdl, ref, E.make_params(first_read=2 * n + r, n_reads=1, **ku), bytes_per_read=2_000_000, events_per_read=600_000)
            assert _record_slice(eng, b, reads, r, nu) == exp["records"].tobytes(), ("unaligned", r)
    finally:
        eng.close()


def test_grch38_like_chimeric(hg002_model):
    """configs[3] on one GPU: 24 linear chromosomes, 3.1 Gb — reference offsets beyond 2^31; chimeric reads join segments of
    different chromosomes (S:1276-1299, 1388-1402)"""
    mdl = hg002_model
    names, bases, off, circ = synth.grch38_like(SEED)
    assert int(off[-1]) > (1 << 31)
    ref = M.Reference(names, bases, off, circ)
    eng = E.Engine(0)
    try:
        eng.set_reference(ref)
        eng.load_model(mdl)
        n = 1_000_000
        kw = dict(seed=SEED + 5, chimeric=True, max_len=min(ref.max_chrom, 1 << 30))
        b = eng.generate(E.make_params(first_read=0, n_reads=n, **kw))
        reads, pieces = b.reads(), b.pieces()
        assert np.all(reads["flags"] == 0)
        al = pieces["kind"] == 0
        clen = np.diff(off.astype(np.int64))
        # every aligned segment inside its chromosome, its global offset consistent with (chrom, pos)
        assert np.all(pieces["pos"][al].astype(np.int64) + pieces["ref_len"][al] <= clen[pieces["chrom"][al]])
        assert np.array_equal(pieces["ref_gpos"][al], off[pieces["chrom"][al]] + pieces["pos"][al])
        assert int((pieces["ref_gpos"][al] > (1 << 31)).sum()) > 100_000                             # chromosomes behind the 2 GB mark are used
        # chromosomes are hit in proportion to their length (uniform start over the concatenated genome, S:1769)
        share = np.bincount(pieces["chrom"][al], weights=pieces["ref_len"][al].astype(np.float64), minlength=24)
        assert np.max(np.abs(share / share.sum() - clen / clen.sum())) < 0.004
        nseg = (reads["n_pieces"].astype(np.int64) + 1) // 2
        assert 0.03 < float((nseg > 1).mean()) < 0.07                                               # segment_mean 1.05
        assert int(reads["n_pieces"].sum()) == len(pieces)
        rec = b.records()
        c_full = checksum(rec)
        rng = np.random.default_rng(3)
        chim = np.nonzero(nseg > 1)[0]
        picks = sorted(set(int(x) for x in rng.integers(0, n, 10)) | set(int(x) for x in chim[:6]) | {0, n - 1})
        for r in picks:
            exp = O.generate(mdl, ref, E.make_params(first_read=r, n_reads=1, **kw), bytes_per_read=2_000_000, events_per_read=200_000)
            got = _record_slice(eng, b, reads, r, n)
            assert got == exp["records"].tobytes(), r
            if nseg[r] > 1:
                name = got.split(b"\n")[0].decode()
                assert "_chimeric_" in name and name.count(";") == 2 * (int(nseg[r]) - 1)
        assert checksum(eng.generate(E.make_params(first_read=0, n_reads=n, **kw)).records()) == c_full
    finally:
        eng.close()


@pytest.mark.parametrize("n,extra,env", [(131_072, dict(emit_errlog=True), {}), (32_768, dict(fastq=True, kmer_bias=5, emit_errlog=True), {}),
                                         (32_768, dict(emit_errlog=True), {"NS_NO_PIECE_THREADS": "1"}),      # one thread per read over the list with a hole
                                         (32_768, dict(emit_errlog=True), {"NS_TAIL_BITS": "31"})])           # ... with the chain tables in global memory
def test_chimeric_batch_on_both_chain_lists_equals_small_batches(hg002_model, monkeypatch, n, extra, env):
    """A chimeric batch large enough for the wave-per-read lists (>= 16 384 reads: the reads of several pieces are visited first, the longest
    of them and the longest 0.1 % of the others on the wave-per-read list, the rest of them as a thread per PIECE — k_chain's piece modes —
    next to the thread-per-read launch of the others) against the SAME reads generated in batches of 4 096 (one thread per read, the path
    the oracle parity tests cover): records and error-profile rows byte for byte, also through the -k stage — a read is a function of
    (seed, index) (S:1276-1299, 1833-1916)"""
    mdl = hg002_model
    bases = synth.synth_sequence(synth.ECOLI_LEN, SEED, n_frac=0.0005, iupac_frac=0.0002, lower_frac=0.02, hp_boost=0.005)
    ref = M.Reference(["ecoli-like"], bases, np.array([0, synth.ECOLI_LEN], dtype=np.uint64), np.array([1], dtype=np.uint8))
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    eng = E.Engine(0)
    try:
        eng.set_reference(ref)
        eng.load_model(mdl)
        chunk = 4_096
        kw = dict(seed=SEED + 11, chimeric=True, max_len=ref.max_chrom, **extra)
        b = eng.generate(E.make_params(first_read=0, n_reads=n, **kw))
        reads = b.reads()
        assert np.all(reads["flags"] == 0)
        nseg = (reads["n_pieces"].astype(np.int64) + 1) // 2
        assert 0.03 < float((nseg > 1).mean()) < 0.07
        rec, err = b.records().copy(), b.errlog().copy()
        seq_len = reads["seq_len"].copy()
        r_at = e_at = 0
        for f in range(0, n, chunk):
            c = eng.generate(E.make_params(first_read=f, n_reads=chunk, **kw))
            assert np.array_equal(c.reads()["seq_len"], seq_len[f:f + chunk]), f
            cr, ce = c.records(), c.errlog()
            assert np.array_equal(cr, rec[r_at:r_at + len(cr)]), f
            assert np.array_equal(ce, err[e_at:e_at + len(ce)]), f
            r_at += len(cr); e_at += len(ce)
        assert r_at == len(rec) and e_at == len(err)
        # ... and reads of several pieces from both sides — the longest (the wave-per-read list's) and three of the middle (a thread per
        # piece) — against the oracle itself
        multi = np.nonzero(nseg > 1)[0]
        by_len = multi[np.argsort(seq_len[multi])]
        for r in (int(x) for x in np.concatenate([by_len[-3:], by_len[len(by_len) // 2:len(by_len) // 2 + 3]])):
            exp = O.generate(mdl, ref, E.make_params(first_read=r, n_reads=1, **kw), bytes_per_read=2_000_000, events_per_read=200_000)
            lo = int(reads["rec_off"][r]); hi = int(reads["rec_off"][r + 1]) if r + 1 < n else len(rec)
            assert rec[lo:hi].tobytes() == exp["records"].tobytes(), r
    finally:
        eng.close()


def test_batches_growing_on_one_engine(hg002_model):
    """Worker calls of growing size on ONE engine, a -k FASTQ call between them (what a CLI run with a changing batch size does): every buffer
    that grows is a new allocation on recycled device memory, and nothing may rely on what an earlier launch left in the old one — the
    slow-tile queue's counter did (zeroed by k_stats_fold in the buffer that the record stage then replaced): this sequence ended in a
    GPU memory fault until the last day of round 6.  The last batch equals the same reads in batches of 4 096."""
    mdl = hg002_model
    bases = synth.synth_sequence(synth.ECOLI_LEN, SEED, n_frac=0.0005, iupac_frac=0.0002, lower_frac=0.02, hp_boost=0.005)
    ref = M.Reference(["ecoli-like"], bases, np.array([0, synth.ECOLI_LEN], dtype=np.uint64), np.array([1], dtype=np.uint8))
    eng = E.Engine(0)
    try:
        eng.set_reference(ref)
        eng.load_model(mdl)
        kw = dict(seed=SEED + 100, chimeric=True, max_len=ref.max_chrom)
        for seed_off in (0, 1):
            for n, extra in ((16_384, dict(emit_errlog=True)), (40_000, dict(emit_errlog=True, fastq=True, kmer_bias=5)), (70_000, dict(emit_errlog=True))):
                b = eng.generate(E.make_params(first_read=7 + seed_off, n_reads=n, **kw, **extra))
                assert np.all(b.reads()["flags"] == 0)
                for f in range(0, n, 4096):                    # (the small calls in between, as scripts/parity_chimeric_big.py makes them)
                    eng.generate(E.make_params(first_read=7 + seed_off + f, n_reads=min(4096, n - f), **kw, **extra))
        n = 70_000
        b = eng.generate(E.make_params(first_read=7, n_reads=n, emit_errlog=True, **kw))
        rec = b.records().copy()
        at = 0
        for f in range(0, n, 4096):
            cr = eng.generate(E.make_params(first_read=7 + f, n_reads=min(4096, n - f), emit_errlog=True, **kw)).records()
            assert np.array_equal(cr, rec[at:at + len(cr)]), f
            at += len(cr)
        assert at == len(rec)
    finally:
        eng.close()


def test_reference_beyond_4gb(hg002_model):
    """three linear chromosomes of 1.6 Gb: the third lies behind offset 2^32 of the concatenated reference (64-bit offsets everywhere)"""
    mdl = hg002_model
    clen = 1_600_000_000
    chunks = [synth.synth_sequence(clen, SEED + 50 + i, iupac_frac=0.0002, hp_boost=0.002) for i in range(3)]
    off = np.array([0, clen, 2 * clen, 3 * clen], dtype=np.uint64)
    ref = M.Reference(["big1", "big2", "big3"], np.concatenate(chunks), off, np.zeros(3, dtype=np.uint8))
    del chunks
    assert int(off[2]) < (1 << 32) < int(off[3])
    eng = E.Engine(0)
    try:
        eng.set_reference(ref)
        eng.load_model(mdl)
        n = 300_000
        kw = dict(seed=SEED + 9, fastq=True, kmer_bias=5, max_len=min(ref.max_chrom, 1 << 30))
        b = eng.generate(E.make_params(first_read=0, n_reads=n, **kw))
        reads, pieces = b.reads(), b.pieces()
        assert np.all(reads["flags"] == 0)
        assert np.array_equal(pieces["ref_gpos"], off[pieces["chrom"]] + pieces["pos"])
        far = np.nonzero(pieces["ref_gpos"] > (1 << 32))[0]
        assert len(far) > 20_000 and np.all(pieces["chrom"][far] == 2)                              # (10 % of the 4.8 Gb lie behind 2^32)
        for r in [int(x) for x in far[:6]] + [0, n - 1]:
            exp = O.generate(mdl, ref, E.make_params(first_read=r, n_reads=1, **kw), bytes_per_read=1_000_000, events_per_read=100_000)
            assert _record_slice(eng, b, reads, r, n) == exp["records"].tobytes(), r
    finally:
        eng.close()


def test_zymo10_like_metagenome(hg002_model):
    """configs[4] on one GPU: one metagenome worker of 10^6 reads on the even community (properties) and a 30 000-read worker on the
    log-distributed community, chimeric, byte for byte against the oracle"""
    mdl = hg002_model
    names, bases, off, circ, species, sp_off, keys, abuns = synth.zymo10_like(SEED)
    mref = MG.MetaReference(M.Reference(names, bases, off, circ), species, sp_off, keys)
    eng = E.Engine(0)
    try:
        eng.set_metagenome(mref, abuns[0], None)
        eng.load_model(mdl)
        n = 1_000_000
        b = eng.generate(E.make_params(seed=SEED, first_read=0, n_reads=n, meta=True, max_len=mref.max_chrom))
        reads, pieces = b.reads(), b.pieces()
        assert len(reads) == n and np.all(reads["flags"] == 0)
        sp_of_chrom = np.searchsorted(sp_off, np.arange(len(names)), side="right") - 1
        al = pieces["kind"] == 0
        per_species = np.bincount(sp_of_chrom[pieces["chrom"][al]], weights=pieces["ref_len"][al].astype(np.float64), minlength=len(species))
        want = np.array([abuns[0][sp] for sp in species])
        assert np.max(np.abs(per_species / per_species.sum() - want / want.sum())) < 0.003            # quotas of assign_species (S:772-775)
        assert np.allclose(eng.species_bases(), per_species)
        clen = np.diff(off.astype(np.int64))
        lin = al & (circ[pieces["chrom"]] == 0)
        assert np.all(pieces["pos"][lin].astype(np.int64) + pieces["ref_len"][lin] <= clen[pieces["chrom"][lin]])
        first = b.records()[:200].tobytes().split(b"\n")[0].decode()
        assert first.startswith(">") and "_aligned_0_" in first and any(first[1:].startswith(sp + "-") for sp in species)
        # the log-distributed sample, chimeric: a whole worker batch equals the oracle's
        infl = {sp: MG.inflate_abun(abuns[1], sp, mdl.abun_inflation) for sp in abuns[1]}
        eng.set_abundance(mref, abuns[1], infl)
        p = E.make_params(seed=SEED + 1, first_read=5_000_000, n_reads=30_000, meta=True, chimeric=True, fastq=True, emit_errlog=True, max_len=mref.max_chrom)
        g = eng.generate(p)
        exp = O.generate_meta(mdl, mref, abuns[1], infl, p, bytes_per_read=60000)
        assert g.records().tobytes() == exp["records"].tobytes()
        assert g.errlog().tobytes() == exp["errlog"].tobytes()
        assert np.allclose(eng.species_bases(), exp["species_bases"])
        top = species[int(np.argmax(exp["species_bases"]))]
        assert top == "Listeria_monocytogenes"                                                       # 89.1 % of the log sample
    finally:
        eng.close()
