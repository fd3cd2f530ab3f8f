"""Pins oracle/ns_oracle.c (the CPU restatement) and the host loader against fixtures produced by the
REAL reference (tests/golden/make_golden.py).  CPU only."""
import ctypes as C
import math

import numpy as np
import pytest

from nanosim_amd import model as M
from tests import oracle_lib as O

TYPE_IX = {"mis": 0, "ins": 1, "del": 2}


def expected_events(e_dict):
    """reference e_dict (sorted by float key) -> [(ceil(key), type, len)] (mutate_read uses ceil, S:1962)."""
    return [(int(math.ceil(k)), TYPE_IX[t], n) for k, t, n in e_dict]


def events_array(ev):
    a = np.zeros(len(ev), dtype=M.EVENT_DTYPE)
    shift = 0
    for i, (p, t, n) in enumerate(ev):
        a[i] = (p, M.ev_pack(n, t, shift))
        shift += n if t == 1 else -n if t == 2 else 0
    return a


def decode_events(ev):
    """[(pos, type, len)] + check that the packed shift is the running sum(ins - del)."""
    out, shift = [], 0
    for e in ev:
        ty, n = int(M.ev_type(e["info"])), int(M.ev_len(e["info"]))
        assert int(M.ev_shift(e["info"])) == shift
        shift += n if ty == 1 else -n if ty == 2 else 0
        out.append((int(e["pos"]), ty, n))
    return out


# ---------------------------------------------------------------------------------------------------
def test_philox_known_answers():
    """Random123 known-answer vectors for philox4x32-10."""
    L = O.lib()
    out = (C.c_uint32 * 4)()
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, exp in kat:
        L.nso_philox(key[0], key[1], ctr[0], ctr[1], ctr[2], ctr[3], out)
        assert tuple(out) == exp


def test_exact_math_accuracy():
    L = O.lib()
    rng = np.random.default_rng(1)
    for x in np.concatenate([10.0 ** rng.uniform(-12, 12, 2000), [1.0, 2.0, 0.5, 1.4142135623730951]]):
        assert abs(L.nso_log(x) - math.log(x)) <= 4e-16 * max(1.0, abs(math.log(x)))
    for y in rng.uniform(-50, 50, 2000):
        assert abs(L.nso_exp(y) / math.exp(y) - 1.0) <= 1e-15
    from scipy.special import ndtri
    ps = np.concatenate([rng.uniform(0, 1, 5000), [1e-10, 1 - 1e-10, 0.02425, 0.97575, 0.5]])
    for p in ps:
        z = L.nso_norminv(p)
        assert abs(z - ndtri(p)) <= 2e-9 * max(1.0, abs(ndtri(p)))
    for x in rng.uniform(0, 5, 200):
        assert abs(L.nso_pow10m1(x) - (10.0 ** x - 1.0)) <= 1e-13 * 10.0 ** x


def test_read_ecdf_matches_reference(golden_functions, small_model):
    for name, cols in (("_first_match.hist", [small_model.first_match]), ("_match_markov_model", small_model.match_markov)):
        fx = golden_functions["ecdf"][name]
        if name == "_first_match.hist":
            fx = fx[:1]
        assert len(fx) == len(cols)
        for f, c in zip(fx, cols):
            assert tuple(f["bin"]) == (c.lo, c.hi_bin)
            segs = c.segments()
            assert len(segs) == len(f["segs"])
            for a, b in zip(segs, f["segs"]):
                assert a == tuple(b)


def _ecdf_ref_lookup(segs, p):
    """the reference's own look-up loop (S:1845-1849) on the fixture's segment list"""
    for lo, hi, vlo, vhi in segs:
        if lo < p <= hi:
            return int(np.floor((p - lo) / (hi - lo) * (vhi - vlo) + vlo))
    return None


def test_ecdf_lookup_transform(golden_functions, small_model):
    L = O.lib()
    rng = np.random.default_rng(5)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    cols = [small_model.first_match] + small_model.match_markov
    fxs = golden_functions["ecdf"]["_first_match.hist"][:1] + golden_functions["ecdf"]["_match_markov_model"]
    for c, f in zip(cols, fxs):
        ps = np.concatenate([rng.uniform(0, 1, 3000), c.hi[:-1], np.nextafter(c.hi[:-1], 2), [1e-12]])
        for p in ps:
            exp = _ecdf_ref_lookup(f["segs"], p)
            if exp is None:
                continue
            assert L.nso_ecdf_lookup(dp(c.hi), dp(c.vhi), len(c.hi), c.vlo0, p) == exp


def test_transition_pick(small_model):
    L = O.lib()
    rng = np.random.default_rng(6)
    for s in range(7):
        a, b, c = small_model.trans_raw[s]
        iv = [((0, a), 0), ((a, a + b), 1), ((1 - c, 1), 2)]          # S:493-495
        row = (C.c_double * 3)(*small_model.trans[s])
        for p in np.concatenate([rng.uniform(0, 1, 2000), [0.0, a, a + b, 1 - c]]):
            exp = None
            for (lo, hi), e in iv:                                       # S:1861-1864
                if lo <= p < hi:
                    exp = e
                    break
            if exp is not None:
                assert L.nso_trans_pick(row, p) == exp


def test_error_list_tape_replay(golden_functions, small_model):
    L = O.lib()
    t = small_model.to_c()
    for case in golden_functions["error_list"]:
        d, keep = O.make_tape(case["u"], case["n"])
        ev = np.zeros(len(case["e_dict"]) + 8, dtype=M.EVENT_DTYPE)
        r = O.NsoElist()
        L.nso_error_list(C.byref(t), case["m_ref"], int(case["fastq"]), C.byref(d), 0, 0, ev.ctypes.data, len(ev), C.byref(r))
        assert not d.tape_err and d.i_u == len(case["u"]) and d.i_n == len(case["n"])
        assert (r.l_new, r.middle_ref) == (case["l_new"], case["middle_ref"])
        exp = expected_events(case["e_dict"])
        assert r.n_ev == len(exp)
        assert decode_events(ev[:r.n_ev]) == exp
        if case["fastq"]:
            assert list(r.e_count) == case["e_count"]


def _run_mutate(L, conv, events, d, want_log=True):
    seg = np.frombuffer(conv.encode(), dtype=np.uint8).copy()
    ev = events_array(events)
    out = np.zeros(len(seg) + int(sum(n for _, t, n in events if t == 1)) + 8, dtype=np.uint8)
    cls = np.zeros_like(out)
    rows = (O.NsoLogRow * (len(events) + 1))()
    txt = np.zeros(2 * sum(n for _, _, n in events) + 8, dtype=np.uint8)
    tl = C.c_uint64(0)
    n = L.nso_mutate_read(seg.ctypes.data, len(seg), ev.ctypes.data, len(ev), C.byref(d), 0, 0, out.ctypes.data,
                          cls.ctypes.data, len(out), C.cast(rows, C.c_void_p), txt.ctypes.data, C.byref(tl))
    assert n >= 0
    log = []
    names = ["mis", "ins", "del"]
    for j in range(len(events)):
        r = rows[j]
        log.append([r.pos, names[r.type], r.len, bytes(txt[r.ref_off:r.ref_off + r.len]).decode(),
                    bytes(txt[r.new_off:r.new_off + r.len]).decode()])
    return bytes(out[:n]).decode(), cls[:n].copy(), log


def test_case_convert_and_mutate_read_tape_replay(golden_functions):
    L = O.lib()
    for case in golden_functions["mutate_read"]:
        if case.get("k"):
            continue        # -k filter is exercised in test_homopolymer.py
        if case["u_convert"] or case["read"] != case["converted"]:
            d, keep = O.make_tape(case["u_convert"])
            buf = np.frombuffer(case["read"].encode(), dtype=np.uint8).copy()
            L.nso_case_convert(buf.ctypes.data, len(buf), C.byref(d), 0, 0)
            assert bytes(buf).decode() == case["converted"] and d.i_u == len(case["u_convert"]) and not d.tape_err
        d, keep = O.make_tape(case["u_mutate"])
        out, cls, log = _run_mutate(L, case["converted"], expected_events(case["e_dict"]), d)
        assert not d.tape_err and d.i_u == len(case["u_mutate"])
        assert out == case["out"]
        assert log == case["log"]


def test_quality_classes_match_reference(golden_functions):
    L = O.lib()
    for case in golden_functions["mutate_fastq_classes"]:
        d = O.make_philox(1, 2)
        out, cls, _ = _run_mutate(L, case["converted"], expected_events(case["e_dict"]), d)
        assert len(out) == case["out_len"]
        assert cls.tolist() == case["classes"]


def test_unaligned_error_list_tape_replay_and_structure(golden_functions, small_model):
    L = O.lib()
    t = small_model.to_c()
    for case in golden_functions["unaligned"]:
        d, keep = O.make_tape(case["u"], case["n"])
        ev = np.zeros(3 * len(case["e_dict"]) + 8, dtype=M.EVENT_DTYPE)
        r = O.NsoElist()
        L.nso_unaligned_error_list(C.byref(t), case["m_ref"], C.byref(d), 128, 0, ev.ctypes.data, len(ev), C.byref(r))
        assert not d.tape_err and d.i_u == len(case["u"]) and d.i_n == len(case["n"])
        assert (r.l_new, r.middle_ref) == (case["l_new"], case["middle_ref"])
        events = decode_events(ev[:r.n_ev])
        # events must be ascending and non-overlapping in reference coordinates
        end = 0
        for p, ty, n in events:
            assert p >= end
            end = p + (0 if ty == 1 else n)
        assert end <= case["middle_ref"]
        out, cls, _ = _run_mutate(L, "A" * case["middle_ref"], events, O.make_philox(3, 4))
        assert len(out) == len(case["copied_mask"]) == case["l_new"]
        mask = "".join("1" if c == M.NS_Q_NAMES.index("match") else "0" for c in cls)
        assert mask == case["copied_mask"]


def test_kde_sample_transform(golden_functions, small_model):
    """sklearn KernelDensity.sample == data[floor(u*n)] + bw*g (fixture asserts that identity at creation);
    here: our loader's vectors + the oracle's 10^x-1 reproduce the reference's samples."""
    L = O.lib()
    idx = {"aligned_region": M.NS_KDE_ALIGNED, "ht_length": M.NS_KDE_HT, "ht_ratio": M.NS_KDE_RATIO}
    for name, fx in golden_functions["kde"].items():
        data, bw = small_model.kde[idx[name]]
        assert bw == fx["bw"]
        u, g, x = np.array(fx["u"]), np.array(fx["g"]), np.array(fx["x"])
        i = (u * len(data)).astype(np.int64)
        mine = data[i] + bw * g
        if fx["log"]:
            mine = np.array([L.nso_pow10m1(v) for v in mine])
        assert np.allclose(mine, x, rtol=1e-12, atol=1e-12)


def test_extract_read_walk(golden_functions, small_ref):
    L = O.lib()
    fx = golden_functions["extract"]
    assert small_ref.names == golden_functions["names"]["seq_names"] == list(fx["seq_len"].keys())
    assert small_ref.genome_len == fx["genome_len"]
    for nm, ln in fx["seq_len"].items():
        i = small_ref.names.index(nm)
        assert int(small_ref.chrom_off[i + 1] - small_ref.chrom_off[i]) == ln
    chrom, pos = C.c_uint32(), C.c_uint64()
    for case in fx["cases"]:
        for v in case["draws"][:-1]:
            assert L.nso_extract_walk(small_ref.chrom_off.ctypes.data, len(small_ref.names), v, case["length"],
                                      C.byref(chrom), C.byref(pos)) == -1
        assert L.nso_extract_walk(small_ref.chrom_off.ctypes.data, len(small_ref.names), case["draws"][-1],
                                  case["length"], C.byref(chrom), C.byref(pos)) == 0
        assert "%s_%d" % (small_ref.names[chrom.value], pos.value) == case["name"]
        if case["seq"] is not None:
            s = small_ref.chrom(chrom.value)[pos.value:pos.value + case["length"]]
            assert bytes(s).decode() == case["seq"]


def test_run_length_tables_vs_reference_samplers(golden_samplers, small_model):
    n = golden_samplers["n"]
    for ty, name in enumerate(("mis", "ins", "del")):
        w = small_model.mix_w[ty]
        c0, c1 = small_model.mix_cdf[ty]
        k = 64
        pmf0 = np.diff(np.concatenate([[0], np.pad(c0, (0, max(0, k - len(c0))), constant_values=1.0)[:k]]))
        pmf1 = np.diff(np.concatenate([[0], np.pad(c1, (0, max(0, k - len(c1))), constant_values=1.0)[:k]]))
        pmf = w * pmf0 + (1 - w) * pmf1                # value v = index+1
        hist = np.array(golden_samplers[name], dtype=np.float64)      # index = value
        emp = hist[1:k] / n
        sigma = np.sqrt(np.maximum(pmf[:k - 1] * (1 - pmf[:k - 1]) / n, 1e-12))
        assert hist[0] == 0
        assert np.all(np.abs(emp - pmf[:k - 1]) <= 5 * sigma + 1e-6), name
        # KS distance well inside the 1 % gate
        assert np.max(np.abs(np.cumsum(emp) - np.cumsum(pmf[:k - 1]))) < 0.004


def test_quality_tables_vs_reference(golden_samplers, small_model):
    for name, fx in golden_samplers["quals"].items():
        cls = M.NS_Q_NAMES.index(name)
        exact = np.zeros(128)
        exact[1:93] = fx["pmf_1_92"]
        raw = M.quality_thresholds(*small_model.quals[name])          # 16-bit thresholds of the closed form: exact to half a unit
        assert np.max(np.abs(raw.astype(np.float64) / 65536.0 - np.cumsum(exact))) <= 1.0 / 65536.0 + 1e-12
        # the table the engine and the oracle use: where several thresholds share a 64-wide bucket of the draw (levels of mass
        # < 2^-10 in the tails) all but one sit on the nearer bucket boundary — at most 32/65536 away from the closed form
        snapped = small_model.qual_thr[cls]
        assert np.array_equal(snapped, M.snap_quality_thresholds(raw)) and np.all(np.diff(snapped.astype(np.int64)) >= 0)
        moved = np.nonzero(snapped != raw)[0]
        assert np.max(np.abs(snapped.astype(np.int64) - raw.astype(np.int64))) <= 32
        assert np.all((raw[moved] < 64 * 8) | (raw[moved] > 65536 - 64 * 8)), "only the tails (mass < 1 %) are touched"
        for b in range(1024):
            assert np.sum((snapped > 64 * b) & (snapped <= 64 * b + 63)) <= 1
        thr = snapped.astype(np.float64) / 65536.0
        pmf = np.diff(np.concatenate([[0.0], thr]))         # P(q = j), j = 0..127
        assert np.max(np.abs(np.cumsum(pmf) - np.cumsum(exact))) <= 33.0 / 65536.0
        hist = np.array(fx["hist"], dtype=np.float64)
        n = hist.sum()
        assert np.max(np.abs(np.cumsum(hist) / n - np.cumsum(exact))) < 0.005


def test_hp_parameters_vs_reference(golden_samplers, small_model):
    for length, vals in golden_samplers["get_nd_par"].items():
        L = float(length)
        for base, (mu_i, sg_i) in (("AT", (0, 1)), ("CG", (4, 5))):
            h = small_model.hp[base]
            mu = h["const"] + h["alpha1"] * L + sum(b * max(L - bp, 0.0) for b, bp in zip(h["betas"], h["breakpoints"]))
            sg = h["intercept"] + h["slope"] * L
            assert mu == pytest.approx(vals[mu_i], rel=1e-12) and sg == pytest.approx(vals[sg_i], rel=1e-12)
    assert small_model.hp_mis_rate == golden_samplers["hp_mis_rate"]
