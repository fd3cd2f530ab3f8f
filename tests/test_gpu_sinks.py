"""The output sinks of the engine (include/nanosim_amd.h: ns_sink_*; the worker's out_reads.write / out_error.write,
src/simulator.py:1437-1443, 2006-2008) on the GPU: files written through them while the next batch is generated are, byte for byte,
what ns_copy_out returns for the same batches — and what the oracle produces."""
import os
import subprocess
import sys

import numpy as np
import pytest

from nanosim_amd import engine as E
from nanosim_amd import simulator
from tests import oracle_lib as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture()
def eng(small_model, small_ref, monkeypatch):
    monkeypatch.setenv("NS_IO_SLICE_BYTES", str(192 * 1024))      # many slices per batch, fewer staging slices than slices in flight
    monkeypatch.setenv("NS_IO_SLICES", "3")
    monkeypatch.setenv("NS_IO_THREADS", "4")
    e = E.Engine(0)
    e.set_reference(small_ref)
    e.load_model(small_model)
    yield e
    e.close()


def test_sinks_write_what_copy_out_returns_while_the_next_batch_runs(tmp_path, eng, small_model, small_ref):
    rec_path, err_path = tmp_path / "reads.fastq", tmp_path / "errors"
    fr = os.open(rec_path, os.O_WRONLY | os.O_CREAT, 0o644)
    fe = os.open(err_path, os.O_WRONLY | os.O_CREAT, 0o644)
    sr, se = eng.sink(fr), eng.sink(fe)
    se.put(simulator.ERR_HEADER)
    want_r, want_e = [], [simulator.ERR_HEADER]
    first = 0
    for n in (700, 1, 1300, 0, 900):                     # ragged batches, an empty one in between; five batches over two result slots
        p = E.make_params(seed=99, first_read=first, n_reads=n, fastq=True, chimeric=True, max_len=small_ref.max_chrom, emit_errlog=True)
        b = eng.generate(p)
        sr.write(E.NS_BUF_RECORDS); se.write(E.NS_BUF_ERRLOG)          # queued: returns at once
        exp = O.generate(small_model, small_ref, p)
        want_r.append(exp["records"].tobytes()); want_e.append(exp["errlog"].tobytes())
        if n == 1300:                                    # the synchronous API still sees the last batch while its copies are in flight
            assert b.records().tobytes() == want_r[-1]
        first += n
    assert sr.drain() == sum(map(len, want_r)) and se.drain() == sum(map(len, want_e))
    c = eng.io_counters()
    assert c["bytes"] == sum(map(len, want_r)) + sum(map(len, want_e)) - len(simulator.ERR_HEADER) and c["n_slices"] == 3
    sr.close(); se.close()
    os.close(fr); os.close(fe)
    assert rec_path.read_bytes() == b"".join(want_r)
    assert err_path.read_bytes() == b"".join(want_e)


def test_sink_without_descriptor_and_dev_null(eng, small_ref):
    p = E.make_params(seed=5, first_read=0, n_reads=500, max_len=small_ref.max_chrom)
    b = eng.generate(p)
    drop = eng.sink(-1)
    fd = os.open("/dev/null", os.O_WRONLY)
    null = eng.sink(fd)
    for _ in range(3):
        drop.write(E.NS_BUF_RECORDS); null.write(E.NS_BUF_RECORDS)
    assert drop.drain() == 3 * int(b.info.record_bytes) == null.drain()
    drop.close(); null.close()
    os.close(fd)


def test_a_failed_write_surfaces_as_eio(tmp_path, eng, small_ref):
    """the reference's worker dies on a failed out_reads.write; here drain() / close() raise NS_EIO with the errno text"""
    b = eng.generate(E.make_params(seed=5, first_read=0, n_reads=800, max_len=small_ref.max_chrom))
    fd = os.open(tmp_path / "ro", os.O_RDONLY | os.O_CREAT, 0o644)          # not open for writing: EBADF
    s = eng.sink(fd)
    s.write(E.NS_BUF_RECORDS)
    with pytest.raises(E.EngineError) as ei:
        s.drain()
    assert ei.value.code == E.NS_EIO and "write" in str(ei.value)
    with pytest.raises(E.EngineError):
        s.close()
    os.close(fd)
    ok = eng.sink(os.open("/dev/null", os.O_WRONLY))                          # the engine goes on working
    eng.generate(E.make_params(seed=5, first_read=0, n_reads=10, max_len=small_ref.max_chrom))
    ok.write(E.NS_BUF_RECORDS); ok.drain(); ok.close()
    assert int(b.info.n_reads) == 800


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_ranks(world, argv):
    env = dict(os.environ, NS_DIST_BACKEND="gloo", NS_DEVICE="0", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "-m", "nanosim_amd.simulator"] + argv
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]


def test_eight_ranks_on_one_gpu_genome_chimeric(tmp_path):
    """the 8-GPU layout of BASELINE configs[3] on one GPU (gloo): eight processes, read-index ranges, one broadcast, sub-files appended in
    rank order — the bytes of the 1-rank run"""
    base = ["genome", "-rg", os.path.join(GOLDEN, "genome_small.fa"), "-c", os.path.join(GOLDEN, "model_small", "training"),
            "-n", "4003", "--seed", "31337", "--chimeric"]
    one = str(tmp_path / "w1" / "sim")
    simulator.main(base + ["-o", one])
    out = str(tmp_path / "w8" / "sim")
    _run_ranks(8, base + ["-o", out, "--merge"])
    assert sorted(os.listdir(tmp_path / "w8")) == sorted(os.listdir(tmp_path / "w1"))
    for f in ("_aligned_reads.fasta", "_aligned_error_profile", "_unaligned_reads.fasta"):
        assert open(out + f, "rb").read() == open(one + f, "rb").read(), f


def test_eight_ranks_on_one_gpu_metagenome(tmp_path, small_model):
    """BASELINE configs[4] layout: every rank is one worker of simulation_aligned_metagenome on its read-index range (own species
    quota, S:835) — each rank's part equals the oracle's worker for that range, the file is their concatenation in rank order"""
    from nanosim_amd import metagenome as MG
    from nanosim_amd import model as M
    from nanosim_amd import shard
    meta = os.path.join(GOLDEN, "meta")
    out = str(tmp_path / "mg8" / "sim")
    _run_ranks(8, ["metagenome", "-gl", os.path.join(meta, "genome_list.tsv"), "-a", os.path.join(meta, "abundance.tsv"),
                   "-dl", os.path.join(meta, "dna_type_list.tsv"), "-c", os.path.join(GOLDEN, "model_small", "training"),
                   "-o", out, "--seed", "777", "--chimeric", "--merge"])
    cwd = os.getcwd()
    os.chdir(ROOT)
    try:
        mref = MG.read_metagenome(os.path.join(meta, "genome_list.tsv"), os.path.join(meta, "dna_type_list.tsv"))
    finally:
        os.chdir(cwd)
    numbers, samples = MG.read_abundance(os.path.join(meta, "abundance.tsv"), mref.species)
    mdl = M.load_model(os.path.join(GOLDEN, "model_small", "training"), chimeric=True)
    assert sorted(os.listdir(tmp_path / "mg8")) == sorted("sim_sample%d_%s" % (s, f) for s in range(len(samples))
                                                          for f in ("aligned_error_profile", "aligned_reads.fasta", "unaligned_reads.fasta"))
    first = 0
    for s, abun in enumerate(samples):
        n_al, n_un = mdl.split_counts(numbers[s])
        infl = {sp: MG.inflate_abun(abun, sp, mdl.abun_inflation) for sp in abun}
        recs, errs = [], [simulator.ERR_HEADER]
        for lo, hi in shard.partition(n_al, 8):
            p = E.make_params(seed=777, first_read=first + lo, n_reads=hi - lo, chimeric=True, max_len=mref.max_chrom, emit_errlog=True, meta=True)
            exp = O.generate_meta(mdl, mref, abun, infl, p)
            recs.append(exp["records"].tobytes()); errs.append(exp["errlog"].tobytes())
        base = out + "_sample%d" % s
        assert open(base + "_aligned_reads.fasta", "rb").read() == b"".join(recs)
        assert open(base + "_aligned_error_profile", "rb").read() == b"".join(errs)
        first += n_al + n_un


@pytest.mark.parametrize("flags", [["--chimeric"], ["--fastq", "-hp", "-k", "5"]])
def test_sub_files_of_minus_t_give_the_same_files(tmp_path, monkeypatch, flags):
    """-t K (S:1588-1639): every batch is cut at read boundaries into K sub-files written side by side and appended in order — the final
    files are those of -t 1; NS_KEEP_SUBFILES=1 keeps the sub-files and lists them; 2 ranks x -t 3 the same"""
    monkeypatch.setattr(simulator, "BATCH_READS", 500)           # several batches: sub-files of batch i close while batch i + 1 is copied
    base = ["genome", "-rg", os.path.join(GOLDEN, "genome_small.fa"), "-c", os.path.join(GOLDEN, "model_small", "training"),
            "-n", "1703", "--seed", "99"] + flags
    ext = ".fastq" if "--fastq" in flags else ".fasta"
    tails = ("_aligned_reads" + ext, "_aligned_error_profile", "_unaligned_reads" + ext)
    one = str(tmp_path / "t1" / "sim")
    simulator.main(base + ["-o", one])
    four = str(tmp_path / "t4" / "sim")
    simulator.main(base + ["-o", four, "-t", "4"])
    assert sorted(os.listdir(tmp_path / "t4")) == sorted(os.listdir(tmp_path / "t1"))           # sub-files merged and removed
    for f in tails:
        assert open(four + f, "rb").read() == open(one + f, "rb").read(), f
    monkeypatch.setenv("NS_KEEP_SUBFILES", "1")
    kept = str(tmp_path / "keep" / "sim")
    simulator.main(base + ["-o", kept, "-t", "4"])
    names = os.listdir(tmp_path / "keep")
    assert "sim_aligned_reads0" + ext in names and "sim_error_profile0" in names and "sim_aligned_reads" + ext not in names
    for f in tails:
        listed = open(kept + f + ".subfiles").read().split()
        assert len(listed) >= 4
        assert b"".join(open(x, "rb").read() for x in listed) == open(one + f, "rb").read(), f
        head = open(listed[1], "rb").read(1)
        assert head in (b"", b">", b"@") or f == "_aligned_error_profile"                          # cut at read boundaries
    monkeypatch.delenv("NS_KEEP_SUBFILES")
    out = str(tmp_path / "w2" / "sim")
    _run_ranks(2, base + ["-o", out, "-t", "3", "--merge"])
    assert sorted(os.listdir(tmp_path / "w2")) == sorted(os.listdir(tmp_path / "t1"))
    for f in tails:
        assert open(out + f, "rb").read() == open(one + f, "rb").read(), f


def test_one_rank_process_group_over_rccl(tmp_path):
    """the multi-GPU code path with the REAL backend on the one GPU of the box: a process group of one rank over RCCL ("nccl"), the
    reference broadcast into a HIP tensor, ns_set_reference_device from its pointer — the bytes of the plain run"""
    base = ["genome", "-rg", os.path.join(GOLDEN, "genome_small.fa"), "-c", os.path.join(GOLDEN, "model_small", "training"),
            "-n", "903", "--seed", "777", "--chimeric", "--fastq"]
    one = str(tmp_path / "plain" / "sim")
    simulator.main(base + ["-o", one])
    out = str(tmp_path / "rccl" / "sim")
    env = dict(os.environ, NS_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    env.pop("NS_DIST_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "-m", "nanosim_amd.simulator"] + base + ["-o", out]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    for f in ("_aligned_reads.fastq", "_aligned_error_profile", "_unaligned_reads.fastq"):
        assert open(out + f, "rb").read() == open(one + f, "rb").read(), f
