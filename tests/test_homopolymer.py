"""-k / --KmerBias: the homopolymer filter of mutate_read (S:1920-1947) and mutate_homo (S:618-705), pinned by tape
replay against the reference (fixtures: tests/golden/reference_functions.json["homopolymer"])."""
import ctypes as C

import numpy as np
import pytest

from nanosim_amd import model as M
from tests import oracle_lib as O
from tests.test_oracle_pin import _run_mutate, decode_events, events_array, expected_events


def filter_events(L, conv, events, k):
    seg = np.frombuffer(conv.encode(), dtype=np.uint8).copy()
    ev = events_array(events)
    sh = C.c_int64(0)
    n = L.nso_hp_filter(seg.ctypes.data, len(seg), ev.ctypes.data, len(ev), k, C.byref(sh))
    kept = decode_events(ev[:n])
    assert sh.value == sum(l if t == 1 else -l if t == 2 else 0 for _, t, l in kept)
    return kept


def test_hp_filter_and_mutate_read_match_reference(golden_functions):
    L = O.lib()
    names = ["mis", "ins", "del"]
    for case in golden_functions["homopolymer"]:
        kept = filter_events(L, case["converted"], expected_events(case["e_dict"]), case["k"])
        ref_rows = sorted((r[0], names.index(r[1]), r[2]) for r in case["log"])
        assert sorted(kept) == ref_rows                       # exactly the events the reference kept
        d, keep = O.make_tape(case["u_mutate"])
        out, cls, log = _run_mutate(L, case["converted"], kept, d)
        assert not d.tape_err and d.i_u == len(case["u_mutate"])
        assert out == case["out1"] and log == case["log"] and cls.tolist() == case["classes1"]


def test_survey_hand_case_with_k(golden_functions):
    L = O.lib()
    case = [c for c in golden_functions["mutate_read"] if c.get("k")][0]
    kept = filter_events(L, case["converted"], expected_events(case["e_dict"]), case["k"])
    d, keep = O.make_tape(case["u_mutate"])
    out, cls, log = _run_mutate(L, case["converted"], kept, d)
    assert out == case["out"] and log == case["log"] and len(out) == 30


def test_mutate_homo_tape_replay(golden_functions, small_model):
    L = O.lib()
    t = small_model.to_c()
    n_runs = 0
    for case in golden_functions["homopolymer"]:
        d, keep = O.make_tape(case["u_homo"], z=case["x_runs"])
        seq = np.frombuffer(case["out1"].encode(), dtype=np.uint8).copy()
        q = np.array(case["classes1"], dtype=np.uint8)
        out = np.zeros(2 * len(seq) + 64, dtype=np.uint8)
        oq = np.zeros_like(out)
        n = L.nso_mutate_homo(C.byref(t), seq.ctypes.data, q.ctypes.data, len(seq), case["k"], C.byref(d), 0, 0,
                              out.ctypes.data, oq.ctypes.data, len(out))
        assert n == len(case["out2"])
        assert not d.tape_err and d.i_u == len(case["u_homo"]) and d.i_z == len(case["x_runs"])
        assert bytes(out[:n]).decode() == case["out2"]
        assert oq[:n].tolist() == case["classes2"]
        n_runs += len(case["x_runs"])
    assert n_runs > 100


def test_get_nd_par(golden_samplers, small_model):
    L = O.lib()
    t = small_model.to_c()
    for length, vals in golden_samplers["get_nd_par"].items():
        for base, (mi, si) in ((ord("A"), (0, 1)), (ord("T"), (2, 3)), (ord("C"), (4, 5)), (ord("G"), (6, 7))):
            assert L.nso_hp_mu(C.byref(t), base, int(length)) == pytest.approx(vals[mi], rel=1e-12)
            assert L.nso_hp_sigma(C.byref(t), base, int(length)) == pytest.approx(vals[si], rel=1e-12)
