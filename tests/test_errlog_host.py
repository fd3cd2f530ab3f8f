"""The packed row writer of the error profile (nanosim_amd/csrc/ns_errlog.h: errlog_tail_v3, what -DNS_ERRLOG_V3 builds into k_errlog) —
the DEVICE source, compiled for the host by tests/errlog_host.hip and run in the kernel's order (tails of a block of 64 rows, then the read
names, then the block) — against the error profile the oracle writes (oracle/ns_oracle.c, S:2006-2008), byte for byte."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

from nanosim_amd import engine as E
from nanosim_amd import model as M
from tests import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
pytestmark = pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="hipcc not found")


@pytest.fixture(scope="module")
def host(tmp_path_factory):
    out = os.path.join(str(tmp_path_factory.mktemp("errlog_host")), "errlog_host.so")
    subprocess.check_call([HIPCC if os.path.exists(HIPCC) else "hipcc", "--cuda-host-only", "-x", "hip", "-O2", "-std=c++17", "-ffp-contract=off",
                           "-DNS_HOST_TEST", "-shared", "-fPIC", "-o", out, os.path.join(ROOT, "tests", "errlog_host.hip")], cwd=ROOT,
                          stderr=subprocess.DEVNULL)
    L = C.CDLL(out)
    L.elhost_ins_tail.restype = C.c_int32; L.elhost_ins_tail.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p]
    L.elhost_errlog.restype = C.c_int64
    L.elhost_errlog.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32,
                                C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64]
    return L


def rows_by_the_device_writer(L, ref, p, out):
    rec = out["records"]
    reads = out["reads"]
    name_len = np.zeros(len(reads), np.uint32)
    for i, rd in enumerate(reads):
        if not rd["flags"]:
            o = int(rd["rec_off"])
            name_len[i] = int(np.argmax(rec[o:o + 4096] == 10)) - 1                      # header line without its '>' / '@'
    bases = np.ascontiguousarray(ref.bases)
    chrom_off = np.ascontiguousarray(ref.chrom_off, dtype=np.uint64)
    buf = np.zeros(len(out["errlog"]) + 4096, np.uint8)
    pieces, events = np.ascontiguousarray(out["pieces"]), np.ascontiguousarray(out["events"])
    n = L.elhost_errlog(reads.ctypes.data, len(reads), pieces.ctypes.data, events.ctypes.data, rec.ctypes.data, name_len.ctypes.data, bases.ctypes.data,
                        len(bases), chrom_off.ctypes.data, len(ref.names), int(p.seed), int(p.first_read), buf.ctypes.data, len(buf))
    assert n >= 0, n
    return buf[:n]


@pytest.mark.parametrize("kw", [dict(n_reads=300), dict(n_reads=200, chimeric=True), dict(n_reads=150, fastq=True, first_read=2 ** 33 + 5)])
def test_packed_rows_equal_the_oracle_profile(host, small_model, small_ref, circ_ref, kw):
    for ref in (small_ref, circ_ref):
        args = dict(seed=20260927, first_read=7, max_len=ref.max_chrom, emit_errlog=True)
        args.update(kw)
        p = E.make_params(**args)
        out = O.generate(small_model, ref, p)
        got = rows_by_the_device_writer(host, ref, p, out)
        assert len(out["errlog"]) > 100000
        assert got.tobytes() == out["errlog"].tobytes()


def test_long_runs_wide_positions_and_iupac_codes(host, tmp_path):
    """Events of more than 8 and more than 16 letters (several stores per column, a second letter word), positions of 8-9 digits (chr1-like
    offsets are reached with a long single chromosome), IUPAC codes under substitutions / deletions (case_convert draws, S:743-755)."""
    from nanosim_amd import synth
    spec = synth.SynthModelSpec(n_train=3000, seed=7, aligned_median=2500.0, mis=(3.0, 0.0, 0.3, 0.5), ins=(8.0, 0.9, 0.12, 0.5),
                                dele=(6.0, 0.95, 0.15, 0.5), mm_means=(2.0, 2.5, 3.0, 3.0, 3.5, 3.5, 4.0, 4.0),
                                mm_zero=(0.0, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3), fm_mean=3.0)
    prefix = str(tmp_path / "dense" / "training")
    synth.write_model(prefix, spec, write_pkl=False)
    mdl = M.load_model(prefix)
    fa = str(tmp_path / "one.fa")
    synth.write_fasta(fa, [("chrL", synth.synth_sequence(12_000_000, 5, iupac_frac=0.02, n_frac=0.01))])
    ref = M.read_fasta(fa, "linear")
    p = E.make_params(seed=99, first_read=0, n_reads=120, max_len=ref.max_chrom, emit_errlog=True)
    out = O.generate(mdl, ref, p)
    ev = out["events"]
    assert int(M.ev_len(ev["info"]).max()) > 16 and int(out["pieces"]["pos"].max()) >= 10_000_000
    got = rows_by_the_device_writer(host, ref, p, out)
    assert got.tobytes() == out["errlog"].tobytes()


def test_number_formats_of_the_packed_writer(host):
    """Positions of 1 to 10 digits (the 10^7 / 10^8 switches of the packed decimal), run lengths of 1 to 4 digits, columns that end on a
    multiple of eight letters: the tail of an insertion row against plain string formatting (the letters themselves: the tests above)."""
    rng = np.random.default_rng(4)
    pos = [0, 9, 10, 99, 9999999, 10000000, 99999999, 100000000, 100000001, 999999999, 1000000000, 4294967295] + \
          [int(rng.integers(0, 2 ** 32)) for _ in range(300)] + [int(10 ** rng.uniform(0, 9.6)) for _ in range(300)]
    lens = [1, 2, 7, 8, 9, 15, 16, 17, 24, 99, 100, 999, 1000, 4095] + [int(rng.integers(1, 40)) for _ in range(40)]
    for i, ps in enumerate(pos):
        for ln in (lens if i < 12 else lens[i % len(lens):][:3]):
            buf = np.full(2 * ln + 64, 0xee, np.uint8)
            n = host.elhost_ins_tail(ps, ln, buf.ctypes.data)
            head = ("\t%d\tins\t%d\t" % (ps, ln)).encode() + b"-" * ln + b"\t"
            assert n == len(head) + ln + 1, (ps, ln)
            row = buf[:n].tobytes()
            assert row.startswith(head) and row.endswith(b"\n") and set(row[len(head):-1]) <= set(b"ACGT"), (ps, ln, row[:40])
            assert (buf[n + 8:] == 0xee).all()                                            # at most 7 bytes behind the row are touched
