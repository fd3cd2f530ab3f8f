"""Distribution parity with the REAL reference (north_star gate: KS distance <= 1 %).

tests/golden/reference_distributions.json holds 2048-point quantile summaries of reads produced by the imported
reference (simulation_aligned_genome / simulation_unaligned) on the committed small model: 114 000 aligned + 120 000
unaligned FASTA reads, 114 000 `--fastq -hp -k 5` reads, 114 000 `--chimeric` reads, 11 400 FASTQ reads (quality
histogram).  Every run is large enough for the 1 % gate (noise floor of two 10^5-read samples: ~0.004).  The same
model is run through the CPU oracle (always) and through the HIP engine (-m gpu) and compared metric by metric."""
import numpy as np
import pytest

from nanosim_amd import engine as E
from nanosim_amd import model as M
from tests import oracle_lib as O

KS_GATE = 0.01


def ks_vs_quantiles(sample, quantiles):
    """max |F_sample(v) - F_ref(v)| over the reference's quantile values (F_ref known to 1/len(quantiles))."""
    q = np.asarray(quantiles, dtype=np.float64)
    vals = np.unique(q)
    f_ref = np.searchsorted(q, vals, side="right") / len(q)
    s = np.sort(np.asarray(sample, dtype=np.float64))
    f_mine = np.searchsorted(s, vals, side="right") / len(s)
    f_ref_l = np.searchsorted(q, vals, side="left") / len(q)
    f_mine_l = np.searchsorted(s, vals, side="left") / len(s)
    return float(max(np.max(np.abs(f_mine - f_ref)), np.max(np.abs(f_mine_l - f_ref_l))))


def per_read_metrics(reads, pieces, events):
    """length, head, tail, ref_len, strand and per-type event / base counts for every (non-chimeric) read."""
    n = len(reads)
    p = pieces[reads["piece_off"]]
    cnt = np.zeros((n, 6), dtype=np.int64)
    ev_off, n_ev = p["ev_off"].astype(np.int64), p["n_ev"].astype(np.int64)
    # flat index of every event -> read id
    rid = np.repeat(np.arange(n), n_ev)
    idx = np.concatenate([np.arange(o, o + k) for o, k in zip(ev_off, n_ev)]) if n else np.zeros(0, np.int64)
    info = events["info"][idx]
    ty, ln = M.ev_type(info).astype(np.int64), M.ev_len(info).astype(np.int64)
    for t in range(3):
        sel = ty == t
        cnt[:, t] = np.bincount(rid[sel], minlength=n)
        cnt[:, 3 + t] = np.bincount(rid[sel], weights=ln[sel], minlength=n).astype(np.int64)
    return dict(len=reads["seq_len"], head=reads["head"], tail=reads["tail"], ref_len=p["ref_len"],
                rev=reads["reversed"], counts=cnt)


def check_aligned(met, fx, tag, gate=KS_GATE, mean_tol=0.01):
    rep = {}
    ref_rl, my_rl = float(np.mean(fx["q_ref_len"])), float(np.mean(met["ref_len"]))
    rep["len"] = ks_vs_quantiles(met["len"], fx["q_len"])
    rep["head"] = ks_vs_quantiles(met["head"], fx["q_head"])
    rep["tail"] = ks_vs_quantiles(met["tail"], fx["q_tail"])
    rep["ref_len"] = ks_vs_quantiles(met["ref_len"], fx["q_ref_len"])
    for j, nm in enumerate(("mis_events", "ins_events", "del_events", "mis_bases", "ins_bases", "del_bases")):
        rep[nm] = ks_vs_quantiles(met["counts"][:, j], fx["q_" + nm])
        assert abs(met["counts"][:, j].mean() / fx["mean_" + nm] - 1.0) < mean_tol, (tag, nm)
        # events (bases) per reference base: insensitive to the read-length sampling noise of the two samples
        assert abs((met["counts"][:, j].mean() / my_rl) / (fx["mean_" + nm] / ref_rl) - 1.0) < 0.006, (tag, nm, "rate")
    assert abs(float(np.mean(met["rev"])) - fx["rev_frac"]) < 0.01
    assert abs(float(np.mean(met["len"])) / fx["mean_len"] - 1.0) < mean_tol
    assert abs((float(np.mean(met["len"])) / my_rl) / (fx["mean_len"] / ref_rl) - 1.0) < 0.002
    bad = {k: v for k, v in rep.items() if v > gate}
    assert not bad, "%s: KS distance above 1 %%: %s (all: %s)" % (tag, bad, rep)
    return rep


def oracle_batch(model, ref, **kw):
    p = E.make_params(seed=424242, first_read=0, max_len=ref.max_chrom, **kw)
    out = O.generate(model, ref, p, bytes_per_read=30000)
    return p, out


_POOL_ARGS = None


def _oracle_chunk(i):
    model, ref, per, kw, sizes, what = _POOL_ARGS
    p = E.make_params(seed=424242, first_read=i * per, n_reads=per, max_len=ref.max_chrom, **kw)
    out = O.generate(model, ref, p, bytes_per_read=sizes[0], events_per_read=sizes[1])
    rd = out["reads"]
    if what == "metrics":               # per-read metrics (+ the quality histogram of a FASTQ run) of the chunk: a few MB instead of its records
        met = per_read_metrics(rd, out["pieces"], out["events"])
        qh = qual_hist_from_records(out["records"], rd) if kw.get("fastq") else None
        return {k: np.array(v) for k, v in met.items()}, qh
    used = int((rd["piece_off"].astype(np.int64) + rd["n_pieces"]).max())
    return rd.copy(), out["pieces"][:used].copy()


def _oracle_pool(model, ref, n_reads, chunks, kw, sizes, what):
    import multiprocessing as mp
    import os
    global _POOL_ARGS
    assert n_reads % chunks == 0
    _POOL_ARGS = (model, ref, n_reads // chunks, kw, sizes, what)
    try:
        with mp.get_context("fork").Pool(min(chunks, os.cpu_count() or 1)) as pool:
            return pool.map(_oracle_chunk, range(chunks))
    finally:
        _POOL_ARGS = None


def oracle_batch_parallel(model, ref, n_reads, chunks=8, bytes_per_read=12000, events_per_read=2000, **kw):
    """reads and pieces of ONE oracle run of n_reads read indices, generated as `chunks` index ranges side by side (forked workers: a read is
    a function of (seed, read index), so the ranges ARE the run — same arrays as oracle_batch but for the rebased piece offsets); no
    records, no events kept.  (One call for 10^5 reads spent most of its time faulting in buffers sized for the longest possible read.)"""
    parts = _oracle_pool(model, ref, n_reads, chunks, dict(kw, emit_records=False), (bytes_per_read, events_per_read), "pieces")
    reads, pieces, base = [], [], 0
    for rd, pc in parts:
        rd = rd.copy(); rd["piece_off"] = rd["piece_off"] + base
        base += len(pc)
        reads.append(rd); pieces.append(pc)
    return np.concatenate(reads), np.concatenate(pieces)


def oracle_metrics_parallel(model, ref, n_reads, chunks=8, bytes_per_read=30000, events_per_read=4000, **kw):
    """per_read_metrics (and the quality histogram of a FASTQ run) of one oracle run, computed range by range in forked workers"""
    parts = _oracle_pool(model, ref, n_reads, chunks, dict(kw), (bytes_per_read, events_per_read), "metrics")
    met = {k: np.concatenate([m[k] for m, _ in parts]) for k in parts[0][0]}
    qh = None if parts[0][1] is None else np.sum([q for _, q in parts], axis=0)
    return met, qh


def test_oracle_aligned_distributions_match_reference(golden_distributions, small_model, small_ref):
    fx = golden_distributions["fasta"]
    met, _ = oracle_metrics_parallel(small_model, small_ref, 60000, emit_records=True)
    check_aligned(met, fx, "oracle")


def test_oracle_unaligned_distributions_match_reference(golden_distributions, small_model, small_ref):
    fx = golden_distributions["fasta"]
    r, _ = oracle_batch_parallel(small_model, small_ref, 100000, kind=E.NS_KIND_UNALIGNED)
    assert ks_vs_quantiles(r["seq_len"], fx["q_unaligned_len"]) <= KS_GATE
    assert abs(float(np.mean(r["reversed"])) - fx["unaligned_rev_frac"]) < 0.01


def test_oracle_homopolymer_mode_distributions_match_reference(golden_distributions, small_model, small_ref):
    """--fastq -hp -k 5 (114 000 reference reads): the events that survive the homopolymer filter, the lengths after mutate_homo"""
    fx = golden_distributions["hp"]
    met, h = oracle_metrics_parallel(small_model, small_ref, 60000, fastq=True, kmer_bias=5)
    check_aligned(met, fx, "oracle-hp")
    h = h.astype(np.float64)
    ref_h = np.array(fx["qual_hist"], dtype=np.float64)
    assert np.max(np.abs(np.cumsum(h) / h.sum() - np.cumsum(ref_h) / ref_h.sum())) <= KS_GATE


def chimeric_metrics(reads, pieces, events):
    """segments per read and the bases its gaps contribute, computed as make_golden.py computes them from the reference's files:
    read length - head - tail - sum(segment reference lengths) - inserted + deleted bases of the logged (aligned-segment) events"""
    n = len(reads)
    nseg = (reads["n_pieces"].astype(np.int64) + 1) // 2
    first = reads["piece_off"].astype(np.int64)
    owner = np.repeat(np.arange(n), reads["n_pieces"].astype(np.int64))
    pidx = np.concatenate([np.arange(o, o + k) for o, k in zip(first, reads["n_pieces"].astype(np.int64))])
    pc = pieces[pidx]
    al = pc["kind"] == 0
    refl = np.bincount(owner[al], weights=pc["ref_len"][al].astype(np.float64), minlength=n).astype(np.int64)
    gap_out = np.bincount(owner[~al], weights=pc["out_len"][~al].astype(np.float64), minlength=n).astype(np.int64)
    seg_out = np.bincount(owner[al], weights=pc["out_len"][al].astype(np.float64), minlength=n).astype(np.int64)
    total = reads["seq_len"].astype(np.int64)
    assert np.array_equal(total, reads["head"].astype(np.int64) + reads["tail"] + seg_out + gap_out)
    # (the reference's l_new counts an insertion that a key collision later replaces, S:1882: what the files show is seg_out)
    return dict(nseg=nseg, gap=total - reads["head"] - reads["tail"] - seg_out, refl=refl)


def check_chimeric(reads, pieces, events, fx, tag, gap_gate=0.03, gap_mean_tol=0.06):
    cm = chimeric_metrics(reads, pieces, events)
    nbin = max(len(fx["nseg_hist"]), int(cm["nseg"].max()) + 1)
    hist = np.bincount(cm["nseg"], minlength=nbin).astype(np.float64)
    ref_h = np.zeros(nbin); ref_h[:len(fx["nseg_hist"])] = fx["nseg_hist"]
    assert np.max(np.abs(np.cumsum(hist) / hist.sum() - np.cumsum(ref_h) / ref_h.sum())) <= KS_GATE, (tag, hist, ref_h)      # S:1276-1279
    ch = cm["nseg"] > 1
    assert fx["gap_bases_nonchimeric_max_abs"] == 0 and np.all(cm["gap"][~ch] == 0)
    # (the 1.05-segment fixture holds ~5 800 chimeric reads: noise floor 0.02; the dense one > 10^5: the 1 % gate)
    assert ks_vs_quantiles(cm["gap"][ch], fx["q_gap_bases_chimeric"]) <= gap_gate, (tag, ks_vs_quantiles(cm["gap"][ch], fx["q_gap_bases_chimeric"]))
    mine = cm["gap"][ch].sum() / (cm["nseg"][ch] - 1).sum()
    assert abs(mine / fx["mean_gap_bases_per_gap"] - 1.0) < gap_mean_tol, (tag, mine, fx["mean_gap_bases_per_gap"])
    assert ks_vs_quantiles(reads["seq_len"], fx["q_len"]) <= KS_GATE, tag
    assert ks_vs_quantiles(cm["refl"], fx["q_ref_len"]) <= KS_GATE, tag
    assert ks_vs_quantiles(reads["head"], fx["q_head"]) <= KS_GATE and ks_vs_quantiles(reads["tail"], fx["q_tail"]) <= KS_GATE, tag
    assert abs(float(np.mean(reads["reversed"])) - fx["rev_frac"]) < 0.01


def test_oracle_genome_chimeric_matches_reference(golden_distributions, small_model, small_ref):
    """genome mode --chimeric (S:1276-1279, 1355-1358, 1406-1419, simulation_gap S:1552-1568): segments per read, gap bases, lengths"""
    import re
    fx = golden_distributions["chimeric"]
    reads, pieces = oracle_batch_parallel(small_model, small_ref, 100000, chimeric=True)
    check_chimeric(reads, pieces, None, fx, "oracle-chimeric")
    # name grammar of a chimeric read (S:1390-1402): ';'-joined positions, _chimeric tag, ';'-joined segment lengths
    name_re = re.compile(r"^[A-Za-z0-9\-]+_\d+(;[A-Za-z0-9\-]+_\d+)+_aligned_\d+_chimeric_[FR]_\d+_\d+(;\d+)+_\d+$")
    assert fx["first"]["chimeric_names"]
    for nm in fx["first"]["chimeric_names"]:
        assert name_re.match(nm), nm
    p, out = oracle_batch(small_model, small_ref, n_reads=400, chimeric=True)
    lines = out["records"].tobytes().decode().split("\n")
    seen = 0
    for i in range(0, len(lines) - 1, 2):
        nm = lines[i][1:]
        if "_chimeric_" in nm:
            assert name_re.match(nm), nm
            parts = nm.split("_")
            assert nm.count(";") == 2 * (len(parts[-2].split(";")) - 1)
            seen += 1
        else:
            assert ";" not in nm
    assert seen > 5


def dense_chimeric_model(small_model, fx):
    """the committed small model with the segment mean of tests/golden/reference_chimeric_dense.json (its only difference: the line of
    <prefix>_chimeric_info that S:571-575 reads; make_golden.py --only-chimeric-dense)"""
    import copy
    m = copy.deepcopy(small_model)
    m.segment_mean = float(fx["segment_mean"])
    m.nseg_cdf = M.geometric_cdf(1.0 / m.segment_mean, 0)                 # np.random.geometric(1 / segment_mean), S:1277
    return m


@pytest.fixture(scope="module")
def golden_chimeric_dense():
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_chimeric_dense.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def golden_chimeric_sparse():
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_chimeric_sparse.json")) as f:
        return json.load(f)


def test_oracle_sparse_chimeric_matches_reference_at_the_1_percent_gate(golden_chimeric_sparse, small_model, small_ref):
    """genome mode --chimeric with the model AS IT IS (1.05 segments per read — the share of the bench model): 1.38 million reference reads,
    > 6 x 10^4 of them chimeric (make_golden.py --only-chimeric-sparse), so the gap bases per chimeric read and the mean gap are held at
    the 1 % gate here too (the 114 000-read run above has 5 578 chimeric reads: noise floor 0.02, gated at 0.03 / 6 %)"""
    fx = golden_chimeric_sparse
    assert sum(fx["nseg_hist"][2:]) > 60000
    reads, pieces = oracle_batch_parallel(small_model, small_ref, 1000000, chunks=40, chimeric=True)
    check_chimeric(reads, pieces, None, fx, "oracle-chimeric-sparse", gap_gate=KS_GATE, gap_mean_tol=0.01)


def test_oracle_dense_chimeric_matches_reference_at_the_1_percent_gate(golden_chimeric_dense, small_model, small_ref):
    """genome mode --chimeric with 2 segments per read on average: 247 000 reference reads, 123 603 of them chimeric — segment counts,
    gap bases per chimeric read (simulation_gap, S:1552-1568) and lengths at KS <= 1 % (the 1.05-segment fixture above has too few
    chimeric reads for that gate)"""
    fx = golden_chimeric_dense
    assert sum(fx["nseg_hist"][2:]) > 100000
    m = dense_chimeric_model(small_model, fx)
    reads, pieces = oracle_batch_parallel(m, small_ref, 120000, bytes_per_read=30000, events_per_read=4000, chimeric=True)
    check_chimeric(reads, pieces, None, fx, "oracle-chimeric-dense", gap_gate=KS_GATE, gap_mean_tol=0.01)


def qual_hist_from_records(records, reads, name_len_total=None):
    """quality histogram over all bases of a FASTQ image"""
    lines = records.tobytes().split(b"\n")
    h = np.zeros(128, dtype=np.int64)
    for i in range(3, len(lines), 4):
        h += np.bincount(np.frombuffer(lines[i], dtype=np.uint8) - 33, minlength=128)[:128]
    return h


def test_oracle_quality_distribution_matches_reference(golden_distributions, small_model, small_ref):
    fx = golden_distributions["fastq"]
    p, out = oracle_batch(small_model, small_ref, n_reads=3000, fastq=True)
    h = qual_hist_from_records(out["records"], out["reads"]).astype(np.float64)
    ref_h = np.array(fx["qual_hist"], dtype=np.float64)
    d = np.max(np.abs(np.cumsum(h) / h.sum() - np.cumsum(ref_h) / ref_h.sum()))
    assert d <= KS_GATE, d


def test_record_grammar_matches_reference(golden_distributions, small_model, small_ref):
    """names / record framing / error-profile rows have the reference's grammar (fixture: first reference records)"""
    import re
    fx = golden_distributions["fasta"]["first"]
    name_re = re.compile(r"^[A-Za-z0-9\-]+_\d+_aligned_\d+_[FR]_\d+_\d+_\d+$")
    for nm in fx["aligned"]:
        assert name_re.match(nm), nm
    p, out = oracle_batch(small_model, small_ref, n_reads=50, emit_errlog=True)
    lines = out["records"].tobytes().decode().split("\n")
    assert lines[-1] == ""
    for i in range(0, len(lines) - 1, 2):
        assert lines[i][0] == ">" and name_re.match(lines[i][1:]), lines[i]
        assert set(lines[i + 1]) <= set("ACGT")
        parts = lines[i][1:].split("_")
        assert len(lines[i + 1]) == out["reads"]["seq_len"][i // 2]
        assert int(parts[-3]) == out["reads"]["head"][i // 2] and int(parts[-1]) == out["reads"]["tail"][i // 2]
    row_re = re.compile(r"^[^\t]+\t\d+\t(mis|ins|del)\t\d+\t[ACGT\-]+\t[ACGT\-]+$")
    for ref_row in fx["err_rows"]:
        if ref_row:
            assert row_re.match(ref_row)
    rows = out["errlog"].tobytes().decode().split("\n")
    assert rows[-1] == ""
    for row in rows[:-1]:
        assert row_re.match(row), row
        f = row.split("\t")
        assert len(f[4]) == len(f[5]) == int(f[3])


@pytest.mark.gpu
def test_gpu_distributions_match_reference(golden_distributions, small_model, small_ref):
    eng = E.Engine(0)
    try:
        eng.set_reference(small_ref)
        eng.load_model(small_model)
        p = E.make_params(seed=99991, first_read=0, n_reads=400000, max_len=small_ref.max_chrom, emit_records=False)
        b = eng.generate(p)
        rep = check_aligned(per_read_metrics(b.reads(), b.pieces(), b.events()), golden_distributions["fasta"], "gpu")
        print("KS distances GPU vs reference:", rep)
        p = E.make_params(seed=99991, first_read=400000, n_reads=100000, kind=E.NS_KIND_UNALIGNED,
                          max_len=small_ref.max_chrom, emit_records=False)
        r = eng.generate(p).reads()
        assert ks_vs_quantiles(r["seq_len"], golden_distributions["fasta"]["q_unaligned_len"]) <= KS_GATE
        p = E.make_params(seed=5, first_read=0, n_reads=20000, fastq=True, max_len=small_ref.max_chrom)
        b = eng.generate(p)
        h = qual_hist_from_records(b.records(), b.reads()).astype(np.float64)
        ref_h = np.array(golden_distributions["fastq"]["qual_hist"], dtype=np.float64)
        assert np.max(np.abs(np.cumsum(h) / h.sum() - np.cumsum(ref_h) / ref_h.sum())) <= KS_GATE
    finally:
        eng.close()


@pytest.mark.gpu
def test_gpu_homopolymer_mode_distributions_match_reference(golden_distributions, small_model, small_ref):
    eng = E.Engine(0)
    try:
        eng.set_reference(small_ref)
        eng.load_model(small_model)
        p = E.make_params(seed=777, first_read=0, n_reads=200000, fastq=True, kmer_bias=5, max_len=small_ref.max_chrom)
        b = eng.generate(p)
        fx = golden_distributions["hp"]
        rep = check_aligned(per_read_metrics(b.reads(), b.pieces(), b.events()), fx, "gpu-hp")
        print("KS distances GPU (-k 5) vs reference:", rep)
    finally:
        eng.close()


@pytest.mark.gpu
def test_gpu_genome_chimeric_matches_reference(golden_distributions, small_model, small_ref):
    eng = E.Engine(0)
    try:
        eng.set_reference(small_ref)
        eng.load_model(small_model)
        p = E.make_params(seed=31337, first_read=0, n_reads=300000, chimeric=True, max_len=small_ref.max_chrom, emit_records=False)
        b = eng.generate(p)
        check_chimeric(b.reads(), b.pieces(), b.events(), golden_distributions["chimeric"], "gpu-chimeric")
    finally:
        eng.close()


@pytest.mark.gpu
def test_gpu_sparse_chimeric_matches_reference_at_the_1_percent_gate(golden_chimeric_sparse, small_model, small_ref):
    """the 1.05-segment model on 3 million reads (~140 000 chimeric) against the 1.38-million-read reference run: gap bases at the 1 % gate"""
    eng = E.Engine(0)
    try:
        eng.set_reference(small_ref)
        eng.load_model(small_model)
        p = E.make_params(seed=161803, first_read=0, n_reads=3000000, chimeric=True, max_len=small_ref.max_chrom, emit_records=False)
        b = eng.generate(p)
        check_chimeric(b.reads(), b.pieces(), None, golden_chimeric_sparse, "gpu-chimeric-sparse", gap_gate=KS_GATE, gap_mean_tol=0.01)
    finally:
        eng.close()


@pytest.mark.gpu
def test_gpu_dense_chimeric_matches_reference_at_the_1_percent_gate(golden_chimeric_dense, small_model, small_ref):
    eng = E.Engine(0)
    try:
        eng.set_reference(small_ref)
        eng.load_model(dense_chimeric_model(small_model, golden_chimeric_dense))
        p = E.make_params(seed=271828, first_read=0, n_reads=300000, chimeric=True, max_len=small_ref.max_chrom, emit_records=False)
        b = eng.generate(p)
        check_chimeric(b.reads(), b.pieces(), b.events(), golden_chimeric_dense, "gpu-chimeric-dense", gap_gate=KS_GATE, gap_mean_tol=0.01)
    finally:
        eng.close()


# ---- the north-star gate on the north-star workload ---------------------------------------------------------------------------------
# tests/golden/reference_hg002.json (make_golden.py --only-hg002): 114 000 aligned + 100 000 unaligned reads of the UNMODIFIED reference
# on the model and the reference bench.py times — `hg002_like` (n_train 10^6 per KDE, aligned regions of 8.4 kb on average) on the
# `ecoli_like` 4 641 652 bp circular chromosome.  Neither input is committed: both are rebuilt here from nanosim_amd/synth.py by seed,
# exactly as bench.py and make_golden.py build them.
HG002_SEED = 20260926


@pytest.fixture(scope="module")
def golden_hg002():
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_hg002.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def hg002_inputs(tmp_path_factory):
    import os
    from nanosim_amd import synth
    d = tmp_path_factory.mktemp("hg002")
    prefix = os.path.join(str(d), "hg002_like")
    synth.write_model(prefix, synth.SynthModelSpec(n_train=1_000_000, seed=HG002_SEED), write_pkl=False)
    mdl = M.load_model(prefix)
    bases = synth.synth_sequence(synth.ECOLI_LEN, HG002_SEED, n_frac=0.0005, iupac_frac=0.0002, lower_frac=0.02, hp_boost=0.005)
    ref = M.Reference(["ecoli-like"], bases, np.array([0, synth.ECOLI_LEN], dtype=np.uint64), np.array([1], dtype=np.uint8))
    return mdl, ref


def test_oracle_hg002_scale_distributions_match_reference(golden_hg002, hg002_inputs):
    """read length, head, tail, reference length, per-type error counts and bases of 8.4 kb reads at KS <= 1 % against the reference"""
    mdl, ref = hg002_inputs
    assert golden_hg002["n_aligned"] >= 100000 and golden_hg002["n_unaligned"] >= 100000
    met, _ = oracle_metrics_parallel(mdl, ref, 48000, chunks=16, bytes_per_read=200000, events_per_read=20000, emit_records=True)
    rep = check_aligned(met, golden_hg002, "oracle-hg002")
    print("KS distances oracle vs reference (hg002_like):", rep)
    r, _ = oracle_batch_parallel(mdl, ref, 60000, bytes_per_read=200000, events_per_read=60000, kind=E.NS_KIND_UNALIGNED)
    assert ks_vs_quantiles(r["seq_len"], golden_hg002["q_unaligned_len"]) <= KS_GATE
    assert abs(float(np.mean(r["reversed"])) - golden_hg002["unaligned_rev_frac"]) < 0.01


@pytest.mark.gpu
def test_gpu_hg002_scale_distributions_match_reference(golden_hg002, hg002_inputs):
    """the same on the HIP path: one launch of 200 000 aligned reads of the bench workload, 100 000 unaligned ones"""
    mdl, ref = hg002_inputs
    eng = E.Engine(0)
    try:
        eng.set_reference(ref)
        eng.load_model(mdl)
        p = E.make_params(seed=HG002_SEED, first_read=0, n_reads=200000, max_len=ref.max_chrom, emit_records=False)
        b = eng.generate(p)
        rep = check_aligned(per_read_metrics(b.reads(), b.pieces(), b.events()), golden_hg002, "gpu-hg002")
        print("KS distances GPU vs reference (hg002_like):", rep)
        p = E.make_params(seed=HG002_SEED, first_read=200000, n_reads=100000, kind=E.NS_KIND_UNALIGNED, max_len=ref.max_chrom, emit_records=False)
        r = eng.generate(p).reads()
        assert ks_vs_quantiles(r["seq_len"], golden_hg002["q_unaligned_len"]) <= KS_GATE
    finally:
        eng.close()

