"""Transcriptome mode (without intron retention): host loader and the oracle's transcript pick / 2-D KDE sample
kept until a transcript repeats / polyA restatements pinned against the reference (tests/golden/reference_transcriptome.json: values of
make_cdf, random.choices and extract_read_trx, and 2 x 96 000 reads of simulation_aligned_transcriptome)."""
import json
import os

import numpy as np
import pytest

from nanosim_amd import engine as E
from nanosim_amd import model as M
from nanosim_amd import transcriptome as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TRX = os.path.join(ROOT, "tests", "golden", "trx")


@pytest.fixture(scope="module")
def fx():
    with open(os.path.join(ROOT, "tests", "golden", "reference_transcriptome.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def trx_ref():
    return T.read_transcriptome(os.path.join(TRX, "transcripts.fa"), os.path.join(TRX, "expression.tsv"), os.path.join(TRX, "polya.txt"), "guppy")


def test_loader_matches_read_profile(fx, trx_ref):
    lens = np.diff(trx_ref.ref.chrom_off.astype(np.int64))
    assert len(trx_ref.ref.names) == fx["n_trx"] and int(trx_ref.polya.sum()) == fx["n_polya"]
    for k, v in fx["seq_len"].items():
        assert lens[trx_ref.ref.names.index(k)] == v
    assert [[trx_ref.ref.names[c], int(lens[c])] for c in trx_ref.expr_chrom] == fx["ecdf_length_list"]        # make_cdf order (S:69-97)
    assert np.array_equal(trx_ref.expr_weight, np.array(fx["ecdf_weight_list"]))                             # bit-identical weights
    for k in fx["polya"]:
        assert trx_ref.polya[trx_ref.ref.names.index(k)] == 1
    assert trx_ref.polya_scale == 4.168299657168961
    assert T.read_transcriptome(os.path.join(TRX, "transcripts.fa"), os.path.join(TRX, "expression.tsv"), None, "albacore").polya_scale == 2.409858743694814


def test_transcript_pick_is_random_choices(fx, trx_ref):
    from tests import oracle_lib as O
    x, keep = O.make_trx(trx_ref)
    L = O.lib()
    for u, pk in zip(fx["choices"]["u"], fx["choices"]["picks"]):
        assert trx_ref.ref.names[trx_ref.expr_chrom[L.nso_trx_pick(x, u)]] == pk[0], u
    # the same rule in numpy (what random.choices does: bisect_right over the running sum, clipped to n - 1)
    cum = trx_ref.expr_cum
    for u, pk in zip(fx["choices"]["u"], fx["choices"]["picks"]):
        i = min(int(np.searchsorted(cum, u * cum[-1], side="right")), len(cum) - 1)
        assert trx_ref.ref.names[trx_ref.expr_chrom[i]] == pk[0]


def test_extract_read_trx_rule(fx, trx_ref):
    """S:1683-1691: start uniform in [0, len - length]; the read keeps a polyA tail if its transcript is listed and it ends within
    10 bases of the 3' end"""
    lens = np.diff(trx_ref.ref.chrom_off.astype(np.int64))
    for e in fx["extract_read_trx"]:
        c = trx_ref.ref.names.index(e["key"])
        assert e["randint"][:2] == [0, int(lens[c]) - e["length"]] and e["pos"] == e["randint"][2]
        assert e["retain"] == bool(trx_ref.polya[c] and e["pos"] + e["length"] + 10 >= lens[c])
        assert trx_ref.ref.chrom(c)[e["pos"]:e["pos"] + 12].tobytes().decode() == e["head"]


@pytest.mark.parametrize("name", ["aligned", "perfect"])
def test_oracle_transcriptome_batches_match_reference_runs(fx, trx_ref, name):
    from tests import oracle_lib as O
    from tests.test_distributions import ks_vs_quantiles
    run = fx["runs"][name]
    perfect = name == "perfect"
    mdl = M.load_model(os.path.join(ROOT, "tests", "golden", "model_small", "training"), transcriptome=True, perfect=perfect)
    lens = np.diff(trx_ref.ref.chrom_off.astype(np.int64))
    p = E.make_params(seed=2024, first_read=0, n_reads=96000, max_len=10 ** 9, trx=True, kind=E.NS_KIND_PERFECT if perfect else E.NS_KIND_ALIGNED,
                      uracil=perfect)
    out = O.generate_trx(mdl, trx_ref, p)
    rd, pc, pa = out["reads"], out["pieces"], out["polya"].astype(np.int64)
    tl, mid = lens[pc["chrom"]].astype(np.float64), pc["ref_len"].astype(np.float64)
    # 96 000 reads against 8 reference workers x 12 000.  Every distribution sits inside the 1 % gate of the north star since the
    # oracle restates S:1080-1104 as it is — ONE 2-D KDE sample kept until a transcript repeats (trx_sampled), per block of 1 024 read
    # indices (oracle/ns_oracle.c: trx_block).  (Until round 3 every attempt drew from the conditional KDE on its
    # own: aligned length at KS 0.009-0.012, the dominant transcript's share 1.7 points off.)
    tol = 0.01
    assert ks_vs_quantiles(mid, run["q_mid"]) < tol and ks_vs_quantiles(mid / tl, run["q_frac"]) < tol
    assert ks_vs_quantiles(pc["pos"] / np.maximum(1, tl - mid), run["q_start_frac"]) < tol
    assert ks_vs_quantiles(rd["tail"] + pa, run["q_tailp"]) < tol and ks_vs_quantiles(rd["head"], run["q_head"]) < tol
    assert ks_vs_quantiles(rd["seq_len"], run["q_seq_len"]) < tol
    assert abs(rd["reversed"].mean() - run["frac_rev"]) < 0.01
    reach = (trx_ref.polya[pc["chrom"]] > 0) & (pc["pos"].astype(np.int64) + pc["ref_len"] + 10 >= tl)
    assert abs(reach.mean() - run["frac_reach_end"]) < 0.01
    assert np.all(pa[~reach] == 0) and np.all(pa[reach] >= 2)                   # int(expon(loc=2)) >= 2
    assert ks_vs_quantiles((rd["tail"] + pa)[reach], run["q_tailp_reach"]) < 0.03      # ~9 000 reads
    if perfect:
        assert np.all(rd["head"] == 0) and np.all(rd["tail"] == 0) and np.all(pc["n_ev"] == 0)
        recs = out["records"].tobytes().split(b"\n")
        assert not any(b"T" in s for s in recs[1:400:2]) and any(b"U" in s for s in recs[1:400:2]) and run["has_u"] and not run["has_t"]
    # expression-weighted pick (S:1084) under the sample-until-repeat rule: every transcript's share within 1 % of the batch — also the one
    # that holds 47 % of the expression (measured: 0.1-0.2 %)
    cnt = np.bincount(pc["chrom"], minlength=len(lens))
    for k, v in run["counts"].items():
        c = int(cnt[trx_ref.ref.names.index(k)])
        assert abs(c - v) < 0.01 * run["n"], (k, v, c)
    # names: <transcript>_<start>_aligned|perfect_<index>_<F|R>_<head>_<middle_ref>_<tail + polyA>
    first = out["records"].tobytes().split(b"\n")[0][1:].decode()
    body, _, rest = first.partition("_perfect_" if perfect else "_aligned_")
    f = rest.split("_")
    assert body.rsplit("_", 1)[0] == trx_ref.ref.names[pc["chrom"][0]] and int(body.rsplit("_", 1)[1]) == pc["pos"][0]
    assert (int(f[0]), f[1], int(f[2]), int(f[3]), int(f[4])) == (0, "R" if rd["reversed"][0] else "F", rd["head"][0], pc["ref_len"][0], rd["tail"][0] + pa[0])


def test_oracle_unaligned_transcriptome_reads(trx_ref):
    """simulation_unaligned("transcriptome") (S:1482-1549 with extract_read S:1695-1703): a uniformly drawn transcript that is
    longer than the read"""
    from tests import oracle_lib as O
    mdl = M.load_model(os.path.join(ROOT, "tests", "golden", "model_small", "training"), transcriptome=True)
    lens = np.diff(trx_ref.ref.chrom_off.astype(np.int64))
    p = E.make_params(seed=9, first_read=0, n_reads=3000, min_len=50, max_len=int(lens.max()), trx=True, kind=E.NS_KIND_UNALIGNED)
    out = O.generate_trx(mdl, trx_ref, p)
    pc = out["pieces"]
    assert np.all(pc["ref_len"] < lens[pc["chrom"]]) and np.all(pc["pos"] + pc["ref_len"] <= lens[pc["chrom"]])
    # short reads fit everywhere: transcripts are drawn uniformly, not by length
    short = pc["ref_len"] < lens.min()
    share = np.bincount(pc["chrom"][short], minlength=len(lens)) / max(1, short.sum())
    assert short.sum() >= 50 and share.max() < 0.06 and (share > 0).sum() > 0.3 * min(len(lens), short.sum())
    assert out["records"].tobytes().split(b"\n")[0].decode().split("_unaligned_")[1].startswith("0_")


def test_oracle_transcriptome_reads_do_not_depend_on_the_batch(trx_ref):
    """The sample-until-repeat rule couples the reads of one BLOCK of 1 024 read indices, not of a batch: a batch that starts inside a
    block walks the block from its start (dry runs), so any split of the run gives the same reads."""
    from tests import oracle_lib as O
    mdl = M.load_model(os.path.join(ROOT, "tests", "golden", "model_small", "training"), transcriptome=True, fastq=True)
    kw = dict(seed=77, max_len=10 ** 9, trx=True, fastq=True, emit_errlog=True)
    whole = O.generate_trx(mdl, trx_ref, E.make_params(first_read=0, n_reads=3000, **kw))
    recs, errs = [], []
    for lo, hi in ((0, 700), (700, 1024), (1024, 2500), (2500, 3000)):
        part = O.generate_trx(mdl, trx_ref, E.make_params(first_read=lo, n_reads=hi - lo, **kw))
        recs.append(part["records"].tobytes()); errs.append(part["errlog"].tobytes())
    assert b"".join(recs) == whole["records"].tobytes() and b"".join(errs) == whole["errlog"].tobytes()
    # inside a sample a transcript is used once: the dominant transcript (47 % of the expression) never fills two neighbouring reads
    # from one sample, yet still holds its share
    names = [ln.split(b"_")[0] for ln in whole["records"].tobytes().split(b"\n")[0::4] if ln]
    top = max(set(names), key=names.count)
    assert 0.40 < names.count(top) / len(names) < 0.50


def test_oracle_pick_walk_replays_the_reference_tape():
    """The walk itself (which pick draws a new KDE sample, which pick leaves the inner loop, S:1080-1104), exactly: the reference's worker
    was run with random.choices, get_length_kde(kde_aligned_2d) and select_nearest_kde2d wrapped (tests/golden/make_golden.py
    --only-trx-walk: 4 000 reads, every pick with the length the reference looked up).  The oracle's walk over those picks and look-ups
    must draw a new sample at exactly the reference's picks and accept exactly the reference's picks; where it skips a look-up ("failed
    before under this sample") the reference must have seen the value of the pick the walk relies on."""
    from tests import oracle_lib as O
    with open(os.path.join(ROOT, "tests", "golden", "reference_trx_walk.json")) as f:
        tape = json.load(f)
    picks = np.array(tape["picks"], dtype=np.int64)
    e, y, ref_redraw, ref_accept = picks[:, 0], picks[:, 1], picks[:, 2], picks[:, 3]
    accept, redraw, memo, n_samples = O.trx_walk_tape(e, y, tape["lengths"])
    assert np.array_equal(redraw, ref_redraw) and np.array_equal(accept, ref_accept)
    assert n_samples == tape["n_samples"] and n_samples > 1000               # 1 + the redraws; the fixture is dominated by one transcript
    assert e[accept == 1].tolist() == tape["written"] and len(tape["written"]) == tape["n_reads"]
    skipped = np.flatnonzero(memo >= 0)
    assert len(skipped) > 0 and not accept[skipped].any()
    assert np.array_equal(y[skipped], y[memo[skipped]]) and np.array_equal(e[skipped], e[memo[skipped]])
    # and a walk that forgot the rule is caught: never drawing a new sample accepts a different pick set on this tape
    lens = np.array(tape["lengths"], dtype=np.int64)
    first_y = {}
    naive = np.array([int(first_y.setdefault(int(t), int(v)) < lens[t]) for t, v in zip(e, y)])
    assert not np.array_equal(naive, ref_accept) or not np.array_equal(y, [first_y[int(t)] for t in e])

SIZES = os.path.join(ROOT, "tests", "golden", "reference_transcriptome_sizes.json")


@pytest.mark.parametrize("key,tol,share_tol", [("w1000_aligned", 0.01, 0.01), ("w1000_perfect", 0.01, 0.01), ("w50000_aligned", 0.01, 0.01),
                                                ("w125_aligned", 0.02, 0.01)])
def test_sample_until_repeat_rule_at_other_worker_sizes(trx_ref, key, tol, share_tol):
    """the per-1 024-block restatement of S:1080-1104 was pinned at ONE worker size (12 000 reads per reference worker).
    The reference's worker keeps a 2-D KDE sample of num_simulate = (reads of the worker) points; the restatement one of unbounded size per
    block of 1 024 read indices.  tests/golden/reference_transcriptome_sizes.json (make_golden.py --only-trx-sizes: the REAL worker on the
    committed inputs) holds 96 x 1 000 reads (aligned and --perfect), 8 x 50 000 and 768 x 125.  Measured: at 1 000 and 50 000 reads per
    worker every distribution sits at KS 0.0013-0.005 and every transcript's share within 0.15 % of n — the 1 % gate holds independent of
    the worker size from 1 000 reads up.  At 125 reads per worker it stops holding for the ALIGNED LENGTH (KS 0.011-0.014) and the dominant
    transcript (45.5 % against 44.8 %): a 125-point sample is replaced after fewer picks and its nearest point is coarse — the reference
    itself moves by that much between -n 125 and -n 1 000 per worker; that case is gated at 2 % and the deviation is written here."""
    from tests import oracle_lib as O
    from tests.test_distributions import ks_vs_quantiles
    with open(SIZES) as f:
        run = json.load(f)[key]
    perfect = key.endswith("perfect")
    mdl = M.load_model(os.path.join(ROOT, "tests", "golden", "model_small", "training"), transcriptome=True, perfect=perfect)
    lens = np.diff(trx_ref.ref.chrom_off.astype(np.int64))
    p = E.make_params(seed=77, first_read=0, n_reads=run["n"], max_len=10 ** 9, trx=True, kind=E.NS_KIND_PERFECT if perfect else E.NS_KIND_ALIGNED,
                      uracil=perfect)
    out = O.generate_trx(mdl, trx_ref, p)
    rd, pc, pa = out["reads"], out["pieces"], out["polya"].astype(np.int64)
    tl, mid = lens[pc["chrom"]].astype(np.float64), pc["ref_len"].astype(np.float64)
    ks = dict(mid=ks_vs_quantiles(mid, run["q_mid"]), frac=ks_vs_quantiles(mid / tl, run["q_frac"]),
              start=ks_vs_quantiles(pc["pos"] / np.maximum(1, tl - mid), run["q_start_frac"]),
              tailp=ks_vs_quantiles(rd["tail"] + pa, run["q_tailp"]), head=ks_vs_quantiles(rd["head"], run["q_head"]),
              seq_len=ks_vs_quantiles(rd["seq_len"], run["q_seq_len"]))
    assert max(ks.values()) < tol, (key, ks)
    if run["per_worker"] >= 1000:                          # (the 1 % gate of the north star, at both sizes)
        assert max(ks.values()) < 0.01, (key, ks)
    cnt = np.bincount(pc["chrom"], minlength=len(lens))
    for k, v in run["counts"].items():
        c = int(cnt[trx_ref.ref.names.index(k)])
        assert abs(c - v) < share_tol * run["n"], (key, k, v, c)
    assert abs(rd["reversed"].mean() - run["frac_rev"]) < 0.01
