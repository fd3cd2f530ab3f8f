"""Full-size checks on the GPU (BASELINE.json configs[1]: E. coli-like circular genome, hg002-like model, 10^6 reads)
through size-independent properties: determinism (checksums), independence from batching, structural invariants of
every read, and replay of the event lists (error CIGAR) of sampled reads against the emitted bases."""
import os

import numpy as np
import pytest

from nanosim_amd import engine as E
from nanosim_amd import model as M
from nanosim_amd import synth
from tests import oracle_lib as O

pytestmark = pytest.mark.gpu
SEED = 20260926
N = 1_000_000


@pytest.fixture(scope="module")
def setup(tmp_path_factory):
    d = tmp_path_factory.mktemp("hg002_like")
    prefix = str(d / "training")
    synth.write_model(prefix, synth.SynthModelSpec(n_train=1_000_000, seed=SEED), write_pkl=False)
    mdl = M.load_model(prefix)
    seq = synth.synth_sequence(synth.ECOLI_LEN, SEED, n_frac=0.0005, iupac_frac=0.0002, lower_frac=0.02, hp_boost=0.005)
    ref = M.Reference(["ecoli-like"], seq, np.array([0, len(seq)], dtype=np.uint64), np.array([1], dtype=np.uint8))
    eng = E.Engine(0)
    eng.set_reference(ref)
    eng.load_model(mdl)
    yield eng, mdl, ref
    eng.close()


def checksum(a: np.ndarray) -> tuple:
    n8 = len(a) // 8 * 8
    w = a[:n8].view(np.uint64)
    return (int(np.bitwise_xor.reduce(w)), int(w.sum(dtype=np.uint64)), int(a[n8:].sum()), len(a))


def test_full_size_determinism_and_batch_independence(setup):
    eng, mdl, ref = setup
    p = E.make_params(seed=SEED, first_read=0, n_reads=N, max_len=ref.max_chrom)
    b = eng.generate(p)
    rec = b.records()
    reads = b.reads()
    c_full = checksum(rec)
    assert int(b.info.record_bytes) == len(rec)
    # idempotence: the same call again gives the same bytes
    b2 = eng.generate(p)
    assert checksum(b2.records()) == c_full
    # two half batches reproduce the full batch (a read is a pure function of (seed, read index))
    h = N // 2
    ra = eng.generate(E.make_params(seed=SEED, first_read=0, n_reads=h, max_len=ref.max_chrom)).records()
    assert checksum(ra) == checksum(rec[:int(reads["rec_off"][h])])
    rb = eng.generate(E.make_params(seed=SEED, first_read=h, n_reads=N - h, max_len=ref.max_chrom)).records()
    assert checksum(rb) == checksum(rec[int(reads["rec_off"][h]):])
    # a different seed gives different bytes
    assert checksum(eng.generate(E.make_params(seed=SEED + 1, first_read=0, n_reads=1000, max_len=ref.max_chrom)).records()) != \
        checksum(rec[:int(reads["rec_off"][1000])])


def test_full_size_structure_and_event_replay(setup):
    eng, mdl, ref = setup
    p = E.make_params(seed=SEED, first_read=0, n_reads=N, max_len=ref.max_chrom, min_len=50)
    b = eng.generate(p)
    reads, pieces = b.reads(), b.pieces()
    assert len(reads) == N and len(pieces) == N
    assert np.all(reads["flags"] == 0) and np.all(reads["n_pieces"] == 1)
    assert np.array_equal(reads["piece_off"], np.arange(N, dtype=np.uint32))
    # S:1377-1382, 1429: emitted length = head + segment + tail, inside [min, max]
    assert np.array_equal(reads["seq_len"], reads["head"] + pieces["out_len"] + reads["tail"])
    assert reads["seq_len"].min() >= 50 and reads["seq_len"].max() <= ref.max_chrom
    assert int(b.info.total_bases) == int(reads["seq_len"].astype(np.int64).sum())
    assert int(b.info.total_ref_bases) == int(pieces["ref_len"].astype(np.int64).sum())
    # circular genome: any start position, segments may wrap (S:1752-1760)
    assert pieces["pos"].max() <= ref.genome_len and np.all(pieces["chrom"] == 0)
    # record offsets: names + framing + sequence, back to back
    rec_len = np.diff(np.concatenate([reads["rec_off"], [np.uint64(b.info.record_bytes)]]).astype(np.int64))
    assert np.all(rec_len > reads["seq_len"].astype(np.int64) + 3)
    # strand ratio and mean length of the model
    assert abs(float(reads["reversed"].mean()) - 0.5) < 0.003
    assert 8000 < reads["seq_len"].mean() < 8800
    # ---- replay the event lists of sampled reads against the emitted bases ----
    rng = np.random.default_rng(1)
    bases = O.normalise_bases(ref.bases)
    comp = np.zeros(256, dtype=np.uint8)
    for x, y in zip(b"ACGT", b"TGCA"):
        comp[x] = y
    L = E.load_library()
    n_events_checked = 0
    for r in rng.integers(0, N, 300):
        pc, rd = pieces[r], reads[r]
        ev = np.empty(int(pc["n_ev"]), dtype=M.EVENT_DTYPE)
        if len(ev):
            eng._check(L.ns_copy_out(eng.ctx, E.NS_BUF_EVENTS, ev.ctypes.data, int(pc["ev_off"]) * 8, ev.nbytes))
        lo = int(rd["rec_off"])
        hi = int(reads["rec_off"][r + 1]) if r + 1 < N else int(b.info.record_bytes)
        rec = np.empty(hi - lo, dtype=np.uint8)
        eng._check(L.ns_copy_out(eng.ctx, E.NS_BUF_RECORDS, rec.ctypes.data, lo, rec.nbytes))
        name, seq = rec.tobytes().split(b"\n")[:2]
        f = name.decode()[1:].split("_")
        assert f[0] == "ecoli-like" and int(f[1]) == pc["pos"] and f[2] == "aligned" and int(f[3]) == r
        assert f[4] == ("R" if rd["reversed"] else "F") and (int(f[5]), int(f[6]), int(f[7])) == (rd["head"], pc["ref_len"], rd["tail"])
        s = np.frombuffer(seq, dtype=np.uint8)
        assert len(s) == rd["seq_len"] and np.all(np.isin(s, np.frombuffer(b"ACGT", dtype=np.uint8)))
        if rd["reversed"]:
            s = comp[s[::-1]]
        body = s[int(rd["head"]):int(rd["head"]) + int(pc["out_len"])]
        idx = (int(pc["pos"]) + np.arange(int(pc["ref_len"]))) % ref.genome_len
        seg = bases[idx]
        unamb = np.isin(seg, np.frombuffer(b"ACGT", dtype=np.uint8))
        # walk the events (ascending): copied bases must equal the reference, substituted ones must differ
        x = 0
        o = 0
        shift = 0
        for e in ev:
            pos, ln, ty = int(e["pos"]), int(M.ev_len(e["info"])), int(M.ev_type(e["info"]))
            assert int(M.ev_shift(e["info"])) == shift and pos >= x
            run = pos - x
            m = unamb[x:pos]
            assert np.array_equal(body[o:o + run][m], seg[x:pos][m])
            o += run; x = pos
            if ty == 0:
                m = unamb[x:x + ln]
                assert np.all(body[o:o + ln][m] != seg[x:x + ln][m])
                o += ln; x += ln
            elif ty == 1:
                o += ln; shift += ln
            else:
                x += ln; shift -= ln
            n_events_checked += 1
        run = int(pc["ref_len"]) - x
        m = unamb[x:]
        assert np.array_equal(body[o:o + run][m], seg[x:][m])
        assert o + run == pc["out_len"]
    assert n_events_checked > 50000


def _record_slice(eng, b, reads, r, n):
    L = E.load_library()
    lo = int(reads["rec_off"][r])
    hi = int(reads["rec_off"][r + 1]) if r + 1 < n else int(b.info.record_bytes)
    rec = np.empty(hi - lo, dtype=np.uint8)
    eng._check(L.ns_copy_out(eng.ctx, E.NS_BUF_RECORDS, rec.ctypes.data, lo, rec.nbytes))
    return rec.tobytes()


@pytest.mark.parametrize("cfg", [dict(), dict(fastq=True, kmer_bias=5), dict(chimeric=True, fastq=True)])
def test_full_size_oracle_spot_checks(setup, tmp_path, cfg):
    """10^6 reads in one launch (configs[1]; with FASTQ + -hp -k 5: the record path of configs[2]; chimeric: configs[3]): reads sampled from the big
    batch are byte-identical to the CPU oracle generating the same read index alone (a read is a pure function of the seed and
    its index), FASTQ framing and quality range hold for every sampled read, two launches agree."""
    eng0, mdl0, ref = setup
    eng, mdl = eng0, mdl0
    own = None
    if cfg:
        prefix = str(tmp_path / "training")
        synth.write_model(prefix, synth.SynthModelSpec(n_train=1_000_000, seed=SEED), write_pkl=False)
        mdl = M.load_model(prefix, fastq=True, homopolymer=True, chimeric=True)
        own = eng = E.Engine(0)
        eng.set_reference(ref)
        eng.load_model(mdl)
    try:
        kw = dict(seed=SEED + 7, max_len=ref.max_chrom, **cfg)
        b = eng.generate(E.make_params(first_read=0, n_reads=N, **kw))
        reads = b.reads()
        assert np.all(reads["flags"] == 0)
        rng = np.random.default_rng(5)
        picks = sorted(set(int(x) for x in rng.integers(0, N, 48)) | {0, N - 1})
        got = {r: _record_slice(eng, b, reads, r, N) for r in picks}
        c_full = checksum(b.records())
        for r in picks:
            exp = O.generate(mdl, ref, E.make_params(first_read=r, n_reads=1, **kw), bytes_per_read=1_000_000, events_per_read=100_000)
            assert got[r] == exp["records"].tobytes(), r
            if cfg.get("fastq"):
                name, seq, plus, qual = got[r].split(b"\n")[:4]
                assert name[:1] == b"@" and plus == b"+" and len(seq) == len(qual) == reads["seq_len"][r]
                q = np.frombuffer(qual, dtype=np.uint8)
                assert q.min() >= 33 + 1 and q.max() <= 33 + 93
        assert checksum(eng.generate(E.make_params(first_read=0, n_reads=N, **kw)).records()) == c_full
    finally:
        if own is not None:
            own.close()
