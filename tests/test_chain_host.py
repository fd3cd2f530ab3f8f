"""The engine's thread-per-read event chains (nanosim_amd/csrc/ns_chain.h: chain_error_list on the integer LDS image and on the fp64
tables, chain_unaligned_error_list) and the table packing of ns_load_model (ns_pack.h) — the DEVICE source, compiled for the host by
tests/chain_host.hip — against the oracle's error_list / unaligned_error_list (oracle/ns_oracle.c, pinned on the reference's tapes in
tests/test_oracle_pin.py), event by event.  The GPU runs the same functions inside k_chain; the -m gpu parity tests compare whole batches.
What this file adds on a box without a GPU: the chain arithmetic (integer thresholds, guides, step lists, staging of four events) is checked
in every CPU run, on the models whose tables take each of the look-up paths."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

from nanosim_amd import model as M
from tests import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
pytestmark = pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="hipcc not found")

VARIANT_LDS, VARIANT_FP64, VARIANT_UNALIGNED, VARIANT_UNALIGNED_G, VARIANT_INT_GLOBAL = 0, 1, 2, 3, 4
UNALIGNED = (VARIANT_UNALIGNED, VARIANT_UNALIGNED_G)
ALL = (VARIANT_LDS, VARIANT_FP64, VARIANT_UNALIGNED, VARIANT_UNALIGNED_G, VARIANT_INT_GLOBAL)


def _build(tmp, extra=()):
    out = os.path.join(tmp, "chain_host.so")
    cmd = [HIPCC if os.path.exists(HIPCC) else "hipcc", "--cuda-host-only", "-x", "hip", "-O2", "-std=c++17", "-ffp-contract=off", "-DNS_HOST_TEST",
           *extra, "-shared", "-fPIC", "-o", out, os.path.join(ROOT, "tests", "chain_host.hip")]
    subprocess.check_call(cmd, cwd=ROOT, stderr=subprocess.DEVNULL)
    L = C.CDLL(out)
    L.chost_pack.restype = C.c_void_p; L.chost_pack.argtypes = [C.POINTER(M.NsModelTables)]
    L.chost_free.restype = None; L.chost_free.argtypes = [C.c_void_p]
    L.chost_whole.restype = C.c_int; L.chost_whole.argtypes = [C.c_void_p]
    L.chost_lds_words.restype = C.c_uint32; L.chost_lds_words.argtypes = [C.c_void_p]
    L.chost_tail_bits.restype = C.c_uint32; L.chost_tail_bits.argtypes = [C.c_void_p]
    L.chost_error_list.restype = C.c_int
    L.chost_error_list.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int32, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32] + \
                                  [C.c_void_p] * 6
    return L


@pytest.fixture(scope="module")
def host(tmp_path_factory):
    return _build(str(tmp_path_factory.mktemp("chain_host")))


def _aligned(cap):
    raw = np.zeros(cap + 4, dtype=M.EVENT_DTYPE)
    skip = (-raw.ctypes.data // 8) % 4                         # 32-byte aligned start: the staged path stores groups of four events
    return raw[skip:skip + cap]


def host_list(L, pk, variant, m_ref, seed, read, seg, attempt, cap, staged=0):
    ev = _aligned(cap)
    out = [C.c_int32(), C.c_int32(), C.c_uint32(), C.c_int32(), C.c_int(), C.c_int()]
    rc = L.chost_error_list(pk, variant, staged, m_ref, seed, read, seg, attempt, ev.ctypes.data, cap, *[C.addressof(o) for o in out])
    assert rc == 0
    l_new, middle_ref, n_ev, shift, overflow, rng = [o.value for o in out]
    return dict(l_new=l_new, middle_ref=middle_ref, n_ev=n_ev, shift=shift, overflow=overflow, range=rng, ev=ev)


def oracle_list(t, unaligned, m_ref, seed, read, seg, attempt, cap):
    Lo = O.lib()
    d = O.make_philox(seed, read)
    ev = np.zeros(cap, dtype=M.EVENT_DTYPE)
    r = O.NsoElist()
    if unaligned:
        Lo.nso_unaligned_error_list(C.byref(t), m_ref, C.byref(d), seg, attempt, ev.ctypes.data, cap, C.byref(r))
    else:
        Lo.nso_error_list(C.byref(t), m_ref, 0, C.byref(d), seg, attempt, ev.ctypes.data, cap, C.byref(r))
    return dict(l_new=r.l_new, middle_ref=r.middle_ref, n_ev=r.n_ev, shift=r.shift, overflow=r.overflow, range=r.range, ev=ev)


def same(h, o, what):
    for k in ("l_new", "middle_ref", "n_ev", "shift"):
        assert h[k] == o[k], (what, k, h[k], o[k])
    assert bool(h["overflow"]) == bool(o["overflow"]) and bool(h["range"]) == bool(o["range"]), what
    n = min(h["n_ev"], len(h["ev"]))
    assert h["ev"][:n].tobytes() == o["ev"][:n].tobytes(), what


def sweep(L, mdl, variants, n_cases, seed, lengths):
    t = mdl.to_c()
    pk = L.chost_pack(C.byref(t))
    assert pk
    try:
        rng = np.random.default_rng(seed)
        n_events = 0
        for i in range(n_cases):
            m_ref = int(lengths[i % len(lengths)]) if i < 2 * len(lengths) else int(rng.integers(1, max(lengths) + 1))
            sd, rd = int(rng.integers(0, 2 ** 63)), int(rng.integers(0, 2 ** 40))
            seg, att = int(rng.choice([0, 1, 5, 128, 130])), int(rng.integers(0, 1000))
            cap = 4 * ((2 * m_ref + 64) // 4)
            for v in variants:
                o = oracle_list(t, v in UNALIGNED, m_ref, sd, rd, seg, att, cap)
                same(host_list(L, pk, v, m_ref, sd, rd, seg, att, cap), o, (v, m_ref, sd, rd, seg, att))
                if v != VARIANT_FP64:
                    same(host_list(L, pk, v, m_ref, sd, rd, seg, att, cap, staged=1), o, ("staged", v, m_ref, sd, rd))
                n_events += o["n_ev"]
        return pk, n_events
    except BaseException:
        L.chost_free(pk)
        raise


def test_device_chains_on_the_small_model(host, small_model):
    pk, n_ev = sweep(host, small_model, ALL, 400, 1, (1, 2, 3, 7, 50, 400, 3000, 20000))
    assert host.chost_whole(pk) and host.chost_lds_words(pk) * 8 <= 40 * 1024 and n_ev > 100000       # the LDS image is what the GPU runs here
    host.chost_free(pk)


def test_device_chains_on_models_that_take_the_other_look_up_paths(host, tmp_path):
    """hg002-like tables (the bench model); a trained-model shape with 15 bins and 1 500-row ECDFs (too large for LDS on the GPU: fp64 path
    there, both here); a dense model (an event every ~3 bases, zero-length matches: the dict-key collision of S:1881-1882, state + 3 rows)."""
    from nanosim_amd import synth
    bins = ((0, 1), (1, 2), (2, 3), (3, 5), (5, 7), (7, 10), (10, 14), (14, 19), (19, 25), (25, 33), (33, 45), (45, 60), (60, 90), (90, 150), (150, 1500))
    specs = dict(hg002=synth.SynthModelSpec(n_train=20000, seed=5),
                 big=synth.SynthModelSpec(n_train=3000, seed=99, ecdf_rows=1500, mm_bins=bins, mm_means=tuple(20.0 + 2 * i for i in range(15)),
                                          mm_zero=(0.0,) + (0.02,) * 14, fm_mean=25.0),
                 # matches of hundreds of bases (a low-error model): previous matches >= 256 are the rule — pm_lut's cells of sixteen lengths, with bin
                 # edges inside a cell (700, 1001) and on a cell border (256, 2048)
                 long=synth.SynthModelSpec(n_train=3000, seed=11, ecdf_rows=3000, fm_mean=150.0,
                                           mm_bins=((0, 40), (40, 120), (120, 256), (256, 700), (700, 1001), (1001, 2048), (2048, 3000)),
                                           mm_means=(150.0, 200.0, 260.0, 320.0, 380.0, 430.0, 480.0), mm_zero=(0.0,) + (0.01,) * 6),
                 dense=synth.SynthModelSpec(n_train=3000, seed=7, aligned_median=2500.0, mis=(3.0, 0.0, 0.3, 0.5), ins=(8.0, 0.9, 0.12, 0.5),
                                            dele=(6.0, 0.95, 0.15, 0.5), mm_means=(2.0, 2.5, 3.0, 3.0, 3.5, 3.5, 4.0, 4.0),
                                            mm_zero=(0.0, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3), fm_mean=3.0))
    short = _build(str(tmp_path), extra=("-DNS_PACK_TAIL_BITS=3",))      # prefixes that end where 1/8 of the probability is left
    for name, spec in specs.items():
        prefix = str(tmp_path / name / "training")
        synth.write_model(prefix, spec, write_pkl=False)
        mdl = M.load_model(prefix)
        pk, n_ev = sweep(host, mdl, ALL, 160, 2, (1, 2, 4, 9, 33, 300, 900, 8000) if name != "long" else (5, 700, 9000, 60000))
        assert host.chost_whole(pk) and n_ev > (20000 if name != "long" else 5000), name
        if name == "long":
            assert host.chost_lds_words(pk) * 8 > 44 * 1024      # (even its shortest prefixes do not fit: on the GPU this model reads its image from global memory)
        elif name == "big":
            # round 6: the LDS image holds the HOT PREFIX of every match-length column (15 x 1 500 segments are 180 KB); a draw behind a
            # prefix takes the full column in global memory.  The image fits three workgroups per CU next to their event staging.
            assert 10 <= host.chost_tail_bits(pk) <= 14 and host.chost_lds_words(pk) * 8 <= 45 * 1024
        else:
            assert host.chost_lds_words(pk) * 8 <= 32 * 1024
        host.chost_free(pk)
        # the same with prefixes so short that one draw in eight leaves them: the full-column path under load (T is a COPY of the LDS words
        # only: a read of a prefix table behind its end would be out of bounds)
        pk, n_ev = sweep(short, mdl, (VARIANT_LDS, VARIANT_INT_GLOBAL), 120, 3, (1, 2, 4, 9, 33, 300, 900, 8000) if name != "long" else (5, 700, 9000, 60000))
        assert short.chost_tail_bits(pk) == 3 and n_ev > (10000 if name != "long" else 3000), name
        short.chost_free(pk)


def test_event_capacity_overflow_and_range_flags(host, small_model, tmp_path):
    """A sink that is too small: the chain keeps counting, flags the overflow and never writes behind the capacity (k_chain re-plans the
    batch from the count); a piece whose cumulative shift leaves the 18-bit field of the event record raises `range` (the batch then takes
    the wide-event path) — the same numbers and flags as the oracle (the lists on the LDS image track both flags arithmetically)."""
    from nanosim_amd import synth
    t = small_model.to_c()
    pk = host.chost_pack(C.byref(t))
    try:
        for v in ALL:
            for staged in ((0,) if v == VARIANT_FP64 else (0, 1)):
                full = oracle_list(t, v in UNALIGNED, 5000, 77, 5, 0, 0, 4096)
                assert full["n_ev"] > 40
                cap = 16
                ev = _aligned(cap + 8)
                ev["pos"] = 0xdeadbeef
                out = [C.c_int32(), C.c_int32(), C.c_uint32(), C.c_int32(), C.c_int(), C.c_int()]
                assert host.chost_error_list(pk, v, staged, 5000, 77, 5, 0, 0, ev.ctypes.data, cap, *[C.addressof(o) for o in out]) == 0
                assert out[2].value == full["n_ev"] and out[4].value == 1 and (out[0].value, out[1].value) == (full["l_new"], full["middle_ref"])
                assert ev[:cap].tobytes() == full["ev"][:cap].tobytes() and (ev["pos"][cap:] == 0xdeadbeef).all()
    finally:
        host.chost_free(pk)
    # insertion-heavy tables: the shift of a long piece runs out of the field
    spec = synth.SynthModelSpec(n_train=3000, seed=8, mis=(3.0, 0.0, 0.3, 0.5), ins=(12.0, 0.9, 0.12, 0.5), dele=(1.5, 0.95, 0.15, 0.5),
                                mm_means=(3.0,) * 8, mm_zero=(0.0,) + (0.2,) * 7, fm_mean=3.0)
    prefix = str(tmp_path / "ins" / "training")
    synth.write_model(prefix, spec, write_pkl=False)
    mdl = M.load_model(prefix)
    t2 = mdl.to_c()
    cap = 4 * 150000
    seen = 0
    pk = host.chost_pack(C.byref(t2))
    try:
        for v in ALL:
            m_ref = 400000 if v in UNALIGNED else 800000          # (shift per base: 0.45 unaligned, 0.18 aligned)
            o = oracle_list(t2, v in UNALIGNED, m_ref, 9, 1, 0, 0, cap)
            h = host_list(host, pk, v, m_ref, 9, 1, 0, 0, cap, staged=0 if v == VARIANT_FP64 else 1)
            same(h, o, ("range", v))
            seen += int(bool(o["range"]))
    finally:
        host.chost_free(pk)
    assert seen == len(ALL)                                # every case does leave the field (else the flag was never exercised)
