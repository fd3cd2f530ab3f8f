"""End-to-end drop-in check on the GPU: the CLI writes the reference's file set and the bytes equal the oracle's."""
import os

import numpy as np
import pytest

from nanosim_amd import engine as E
from nanosim_amd import model as M
from nanosim_amd import simulator
from tests import oracle_lib as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.mark.parametrize("fastq", [False, True])
def test_genome_mode_end_to_end(tmp_path, fastq, small_ref):
    out = str(tmp_path / "run" / "simulated")
    argv = ["genome", "-rg", os.path.join(GOLDEN, "genome_small.fa"), "-c", os.path.join(GOLDEN, "model_small", "training"),
            "-o", out, "-n", "400", "--seed", "12345", "--chimeric"] + (["--fastq"] if fastq else [])
    simulator.main(argv)
    ext = ".fastq" if fastq else ".fasta"
    files = sorted(os.listdir(tmp_path / "run"))
    assert files == sorted(["simulated_aligned_error_profile", "simulated_aligned_reads" + ext, "simulated_unaligned_reads" + ext])
    mdl = M.load_model(os.path.join(GOLDEN, "model_small", "training"), chimeric=True, fastq=fastq)
    n_al, n_un = mdl.split_counts(400)
    assert (n_al, n_un) == (380, 20)
    p = E.make_params(seed=12345, first_read=0, n_reads=n_al, fastq=fastq, chimeric=True, max_len=small_ref.max_chrom, emit_errlog=True)
    exp = O.generate(mdl, small_ref, p)
    assert open(out + "_aligned_reads" + ext, "rb").read() == exp["records"].tobytes()
    assert open(out + "_aligned_error_profile", "rb").read() == simulator.ERR_HEADER + exp["errlog"].tobytes()
    p = E.make_params(seed=12345, first_read=n_al, n_reads=n_un, kind=E.NS_KIND_UNALIGNED, fastq=fastq, max_len=small_ref.max_chrom)
    exp = O.generate(mdl, small_ref, p)
    got = open(out + "_unaligned_reads" + ext, "rb").read()
    assert got == exp["records"].tobytes()
    first = got.split(b"\n")[0].decode()
    assert "_unaligned_%d_" % n_al in first          # numbering continues after the aligned reads (S:1506-1508)


def test_perfect_mode(tmp_path, small_ref):
    out = str(tmp_path / "perfect")
    simulator.main(["genome", "-rg", os.path.join(GOLDEN, "genome_small.fa"), "-c", os.path.join(GOLDEN, "model_small", "training"),
                    "-o", out, "-n", "100", "--seed", "5", "--perfect"])
    assert not os.path.exists(out + "_unaligned_reads.fasta")       # S:1642: no unaligned phase with --perfect
    lines = open(out + "_aligned_reads.fasta").read().split("\n")
    assert len(lines) == 201
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    for i in range(0, 200, 2):
        f = lines[i][1:].split("_")
        chrom, pos, strand, ln = f[0], int(f[1]), f[4], int(f[6])
        ci = small_ref.names.index(chrom)
        src = O.normalise_bases(small_ref.chrom(ci)[pos:pos + ln])
        seq = lines[i + 1] if strand == "F" else "".join(comp[c] for c in reversed(lines[i + 1]))
        assert len(seq) == ln
        unamb = np.isin(src, np.frombuffer(b"ACGT", dtype=np.uint8))
        assert np.all(np.frombuffer(seq.encode(), dtype=np.uint8)[unamb] == src[unamb])


def test_metagenome_mode_end_to_end(tmp_path, small_model):
    """Two samples (3000 + 500 reads), chimeric FASTQ: the reference's per-sample file set; every sample equals the oracle's
    worker for the same (seed, read range, abundances)."""
    from nanosim_amd import metagenome as MG
    meta = os.path.join(GOLDEN, "meta")
    out = str(tmp_path / "mg" / "sim")
    cwd = os.getcwd()
    os.chdir(ROOT)                               # the genome list holds paths relative to the repo root
    try:
        simulator.main(["metagenome", "-gl", os.path.join(meta, "genome_list.tsv"), "-a", os.path.join(meta, "abundance.tsv"),
                        "-dl", os.path.join(meta, "dna_type_list.tsv"), "-c", os.path.join(GOLDEN, "model_small", "training"),
                        "-o", out, "--seed", "777", "--chimeric", "--fastq"])
        mref = MG.read_metagenome(os.path.join(meta, "genome_list.tsv"), os.path.join(meta, "dna_type_list.tsv"))
    finally:
        os.chdir(cwd)
    files = sorted(os.listdir(tmp_path / "mg"))
    assert files == sorted("sim_sample%d_%s" % (s, f) for s in (0, 1)
                           for f in ("aligned_error_profile", "aligned_reads.fastq", "unaligned_reads.fastq"))
    numbers, samples = MG.read_abundance(os.path.join(meta, "abundance.tsv"), mref.species)
    mdl = M.load_model(os.path.join(GOLDEN, "model_small", "training"), chimeric=True, fastq=True)
    first = 0
    for s, abun in enumerate(samples):
        n_al, n_un = mdl.split_counts(numbers[s])
        infl = {sp: MG.inflate_abun(abun, sp, mdl.abun_inflation) for sp in abun}
        p = E.make_params(seed=777, first_read=first, n_reads=n_al, fastq=True, chimeric=True, max_len=mref.max_chrom, emit_errlog=True, meta=True)
        exp = O.generate_meta(mdl, mref, abun, infl, p)
        base = out + "_sample%d" % s
        assert open(base + "_aligned_reads.fastq", "rb").read() == exp["records"].tobytes()
        assert open(base + "_aligned_error_profile", "rb").read() == simulator.ERR_HEADER + exp["errlog"].tobytes()
        p = E.make_params(seed=777, first_read=first + n_al, n_reads=n_un, kind=E.NS_KIND_UNALIGNED, fastq=True, max_len=mref.max_chrom, meta=True)
        exp = O.generate_meta(mdl, mref, abun, None, p)
        assert open(base + "_unaligned_reads.fastq", "rb").read() == exp["records"].tobytes()
        first += n_al + n_un


def test_transcriptome_mode_end_to_end(tmp_path):
    """transcriptome --no_model_ir --polya --uracil --fastq: the reference's file set; bytes equal the oracle's"""
    from nanosim_amd import transcriptome as TR
    trx = os.path.join(GOLDEN, "trx")
    out = str(tmp_path / "tx" / "sim")
    argv = ["transcriptome", "-rt", os.path.join(trx, "transcripts.fa"), "-e", os.path.join(trx, "expression.tsv"), "--polya",
            os.path.join(trx, "polya.txt"), "-b", "guppy", "-c", os.path.join(GOLDEN, "model_small", "training"), "-o", out, "-n", "1500",
            "--seed", "4242", "--no_model_ir", "--uracil", "--fastq"]
    simulator.main(argv)
    files = sorted(os.listdir(tmp_path / "tx"))
    assert files == sorted(["sim_aligned_error_profile", "sim_aligned_reads.fastq", "sim_unaligned_reads.fastq"])
    tr = TR.read_transcriptome(os.path.join(trx, "transcripts.fa"), os.path.join(trx, "expression.tsv"), os.path.join(trx, "polya.txt"), "guppy")
    mdl = M.load_model(os.path.join(GOLDEN, "model_small", "training"), transcriptome=True, fastq=True)
    n_al, n_un = mdl.split_counts(1500)
    p = E.make_params(seed=4242, first_read=0, n_reads=n_al, fastq=True, max_len=tr.ref.max_chrom, emit_errlog=True, trx=True, uracil=True)
    exp = O.generate_trx(mdl, tr, p)
    assert open(out + "_aligned_reads.fastq", "rb").read() == exp["records"].tobytes()
    assert open(out + "_aligned_error_profile", "rb").read() == simulator.ERR_HEADER + exp["errlog"].tobytes()
    p = E.make_params(seed=4242, first_read=n_al, n_reads=n_un, kind=E.NS_KIND_UNALIGNED, fastq=True, max_len=tr.ref.max_chrom, trx=True, uracil=True)
    assert open(out + "_unaligned_reads.fastq", "rb").read() == O.generate_trx(mdl, tr, p)["records"].tobytes()
    seqs = open(out + "_aligned_reads.fastq").read().split("\n")[1::4]
    assert not any("T" in x for x in seqs) and any("U" in x for x in seqs)
    # intron retention (the default): needs the genome
    with pytest.raises(SystemExit) as e:
        simulator.main(["transcriptome", "-rt", os.path.join(trx, "transcripts.fa"), "-e", os.path.join(trx, "expression.tsv"), "-c",
                        os.path.join(GOLDEN, "model_small", "training"), "-o", out])
    assert e.value.code == 1


def test_transcriptome_mode_with_intron_retention(tmp_path):
    """transcriptome -rg genome.fa (model_ir on): <prefix>_IR_markov_model + <prefix>_added_intron_final.gff3 are read, reads with a
    retained intron carry it in their name; bytes equal the oracle's"""
    from nanosim_amd import intron_retention as IR
    from nanosim_amd import transcriptome as TR
    trx = os.path.join(GOLDEN, "trx")
    out = str(tmp_path / "ir" / "sim")
    prefix = os.path.join(GOLDEN, "model_small", "training")
    simulator.main(["transcriptome", "-rt", os.path.join(trx, "transcripts.fa"), "-rg", os.path.join(trx, "genome.fa"), "-e",
                    os.path.join(trx, "expression.tsv"), "--polya", os.path.join(trx, "polya.txt"), "-b", "guppy", "-c", prefix, "-o", out,
                    "-n", "1200", "--seed", "99"])
    tr = TR.read_transcriptome(os.path.join(trx, "transcripts.fa"), os.path.join(trx, "expression.tsv"), os.path.join(trx, "polya.txt"), "guppy")
    ir = IR.load(prefix, os.path.join(trx, "genome.fa"), tr.ref)
    tr = TR.restrict_expression(tr, ir.eligible)
    mdl = M.load_model(prefix, transcriptome=True)
    n_al, n_un = mdl.split_counts(1200)
    p = E.make_params(seed=99, first_read=0, n_reads=n_al, max_len=tr.ref.max_chrom, emit_errlog=True, trx=True, model_ir=True)
    exp = O.generate_trx(mdl, tr, p, ir=ir)
    got = open(out + "_aligned_reads.fasta", "rb").read()
    assert got == exp["records"].tobytes()
    assert got.count(b"_RetainedIntron_") > 20
    assert open(out + "_aligned_error_profile", "rb").read() == simulator.ERR_HEADER + exp["errlog"].tobytes()
    p = E.make_params(seed=99, first_read=n_al, n_reads=n_un, kind=E.NS_KIND_UNALIGNED, max_len=tr.ref.max_chrom, trx=True)
    assert open(out + "_unaligned_reads.fasta", "rb").read() == O.generate_trx(mdl, tr, p)["records"].tobytes()


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_ranks(world, argv, tmp_path):
    """the CLI under torch.distributed.run with `world` ranks, ALL on GPU 0 (NS_DEVICE), the reference broadcast over gloo"""
    import subprocess
    import sys
    env = dict(os.environ, NS_DIST_BACKEND="gloo", NS_DEVICE="0", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "-m", "nanosim_amd.simulator"] + argv
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("flags", [["--chimeric"], ["--fastq", "-hp", "-k", "5"]])
def test_genome_output_does_not_depend_on_the_number_of_ranks(tmp_path, flags):
    """a read is a function of (seed, read index) — 2 and 3 ranks (one process per rank, read-index ranges, one
    broadcast of the reference, every rank writing at its final file offsets) produce the bytes of the 1-rank run."""
    base = ["genome", "-rg", os.path.join(GOLDEN, "genome_small.fa"), "-c", os.path.join(GOLDEN, "model_small", "training"),
            "-n", "1203", "--seed", "424242"] + flags
    ext = ".fastq" if "--fastq" in flags else ".fasta"
    one = str(tmp_path / "w1" / "sim")
    simulator.main(base + ["-o", one])
    for world in (2, 3):
        out = str(tmp_path / ("w%d" % world) / "sim")
        _run_ranks(world, base + ["-o", out, "--merge"], tmp_path)                                        # --merge: the reference's single files
        assert sorted(os.listdir(tmp_path / ("w%d" % world))) == sorted(os.listdir(tmp_path / "w1"))      # no sub-files left behind
        for f in ("_aligned_reads" + ext, "_aligned_error_profile", "_unaligned_reads" + ext):
            assert open(out + f, "rb").read() == open(one + f, "rb").read(), (world, f)
    # the default with several ranks (round 6): the parts stay where the ranks wrote them, <file>.subfiles lists them in rank order
    out = str(tmp_path / "w2keep" / "sim")
    _run_ranks(2, base + ["-o", out], tmp_path)
    for f in ("_aligned_reads" + ext, "_aligned_error_profile", "_unaligned_reads" + ext):
        listed = open(out + f + ".subfiles").read().split()
        assert len(listed) == 2 and listed[0] == os.path.abspath(out + f)                                 # rank 0 wrote the head of the final file
        assert b"".join(open(x, "rb").read() for x in listed) == open(one + f, "rb").read(), f


def test_transcriptome_output_does_not_depend_on_the_number_of_ranks(tmp_path):
    trx = os.path.join(GOLDEN, "trx")
    base = ["transcriptome", "-rt", os.path.join(trx, "transcripts.fa"), "-e", os.path.join(trx, "expression.tsv"),
            "-c", os.path.join(GOLDEN, "model_small", "training"), "-n", "900", "--seed", "99", "--no_model_ir", "--fastq"]
    one = str(tmp_path / "w1" / "sim")
    simulator.main(base + ["-o", one])
    out = str(tmp_path / "w2" / "sim")
    _run_ranks(2, base + ["-o", out, "--merge"], tmp_path)
    for f in ("_aligned_reads.fastq", "_aligned_error_profile", "_unaligned_reads.fastq"):
        assert open(out + f, "rb").read() == open(one + f, "rb").read(), f


def test_configs0_ecoli_circular_perfect_10k_reads(tmp_path):
    """BASELINE configs[0] as stated: E. coli-size CIRCULAR genome x --perfect x 10 000 reads x FASTA (S:1321-1343, 1750-1781).  The file
    equals the oracle's bytes, and every read — also those across the origin — is a substring of the doubled genome (or of its reverse
    complement) at the position its name gives."""
    from nanosim_amd import synth
    prefix = str(tmp_path / "model" / "hg002_like")
    os.makedirs(os.path.dirname(prefix))
    synth.write_model(prefix, synth.SynthModelSpec(n_train=200_000, seed=11), write_pkl=False)
    seq = synth.synth_sequence(synth.ECOLI_LEN, 11, n_frac=0.0, iupac_frac=0.0, lower_frac=0.02, hp_boost=0.005)
    fa = str(tmp_path / "ecoli_like.fa")
    synth.write_fasta(fa, [("ecoli-like", seq)])
    out = str(tmp_path / "run" / "simulated")
    simulator.main(["genome", "-rg", fa, "-c", prefix, "-o", out, "-n", "10000", "--seed", "2026", "-dna_type", "circular", "--perfect"])
    assert sorted(os.listdir(tmp_path / "run")) == ["simulated_aligned_error_profile", "simulated_aligned_reads.fasta"]
    assert open(out + "_aligned_error_profile", "rb").read() == simulator.ERR_HEADER           # S:1634: header only
    ref = M.read_fasta(fa, "circular")
    mdl = M.load_model(prefix, perfect=True)
    assert mdl.split_counts(10000) == (10000, 0)                                                # S:465-467
    p = E.make_params(seed=2026, first_read=0, n_reads=10000, kind=E.NS_KIND_PERFECT, max_len=ref.max_chrom, emit_errlog=True)
    exp = O.generate(mdl, ref, p, bytes_per_read=60000)
    got = open(out + "_aligned_reads.fasta", "rb").read()
    assert got == exp["records"].tobytes()
    lines = got.split(b"\n")
    assert len(lines) == 20001
    genome = O.normalise_bases(ref.chrom(0))
    doubled = np.concatenate([genome, genome]).tobytes()
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    wraps = 0
    for i in range(0, 20000, 2):
        f = lines[i][1:].split(b"_")                 # ecoli-like_<pos>_perfect_<n>_<F|R>_0_<len>_0
        assert f[2] == b"perfect" and int(f[3]) == i // 2 and f[5] == b"0" and f[7] == b"0"
        pos, ln = int(f[1]), int(f[6])
        s = lines[i + 1] if f[4] == b"F" else lines[i + 1].translate(comp)[::-1]
        assert len(s) == ln and doubled[pos:pos + ln] == s, lines[i]
        wraps += pos + ln > len(genome)
    assert wraps >= 5                                # reads across the origin exist in this sample (expected ~18)


@pytest.mark.parametrize("mode", ["genome", "metagenome", "transcriptome"])
def test_two_context_schedule_writes_the_files_of_the_serial_one(tmp_path, monkeypatch, mode):
    """the CLI runs the unaligned worker calls on a background engine context NEXT TO the aligned ones (the schedule
    bench.py times); the phase order of S:1588-1672 constrains the FILE CONTENT only — the files equal those of NS_SERIAL=1 (one
    context, aligned then unaligned), byte for byte, over several batches per phase."""
    monkeypatch.setattr(simulator, "BATCH_READS", 700)
    prefix = os.path.join(GOLDEN, "model_small", "training")
    if mode == "genome":
        argv = ["genome", "-rg", os.path.join(GOLDEN, "genome_small.fa"), "-c", prefix, "-n", "3000", "--seed", "8", "--chimeric", "--fastq", "-hp", "-k", "5"]
    elif mode == "metagenome":
        meta = os.path.join(GOLDEN, "meta")
        argv = ["metagenome", "-gl", os.path.join(meta, "genome_list.tsv"), "-a", os.path.join(meta, "abundance.tsv"),
                "-dl", os.path.join(meta, "dna_type_list.tsv"), "-c", prefix, "--seed", "8", "--chimeric"]
    else:
        trx = os.path.join(GOLDEN, "trx")
        argv = ["transcriptome", "-rt", os.path.join(trx, "transcripts.fa"), "-rg", os.path.join(trx, "genome.fa"), "-e",
                os.path.join(trx, "expression.tsv"), "-c", prefix, "-n", "2500", "--seed", "8", "--fastq"]
    cwd = os.getcwd()
    os.chdir(ROOT)                               # (the metagenome genome list holds paths relative to the repo root)
    try:
        simulator.main(argv + ["-o", str(tmp_path / "two" / "sim")])       # default: one ns_generate_step per step (StepPair, round 5)
        monkeypatch.setenv("NS_TWO_ENGINES", "1")                            # two engine contexts of their own + a Python thread (rounds 2-4)
        simulator.main(argv + ["-o", str(tmp_path / "two_engines" / "sim")])
        monkeypatch.delenv("NS_TWO_ENGINES")
        monkeypatch.setenv("NS_SERIAL", "1")
        simulator.main(argv + ["-o", str(tmp_path / "one" / "sim")])
    finally:
        os.chdir(cwd)
    names = sorted(os.listdir(tmp_path / "one"))
    assert names == sorted(os.listdir(tmp_path / "two")) == sorted(os.listdir(tmp_path / "two_engines")) and len(names) >= 3
    for f in names:
        a, b, c = (open(tmp_path / d / f, "rb").read() for d in ("one", "two", "two_engines"))
        assert a == b == c and (len(a) > 0 or "unaligned" in f), f


def test_generate_step_is_two_worker_calls_side_by_side(small_model, small_ref):
    """VERDICT r4 item 2 / ABI 6: ns_generate_step runs the aligned worker call on the context and the unaligned one on its step companion
    (same reference and model, own buffers, the library's worker thread).  Bytes of both batches == two ns_generate calls; either half
    may be missing; the companion's results are reached through Engine.step_engine(); its tables cannot be set behind its owner's back."""
    import numpy as np
    from nanosim_amd import engine as E
    eng, ref_eng = E.Engine(0), E.Engine(0)
    try:
        for e in (eng, ref_eng):
            e.set_reference(small_ref); e.load_model(small_model)
        for it, (n_al, n_un) in enumerate(((3000, 400), (1200, 1300), (257, 0), (0, 300), (3000, 400))):
            p_al = E.make_params(seed=5, first_read=1000 * it, n_reads=n_al, max_len=small_ref.max_chrom, chimeric=True, fastq=True, emit_errlog=True) if n_al else None
            p_un = E.make_params(seed=5, first_read=1000 * it + n_al, n_reads=n_un, kind=E.NS_KIND_UNALIGNED, max_len=small_ref.max_chrom, fastq=True) if n_un else None
            b_al, b_un = eng.generate_step(p_al, p_un)
            assert (b_al is None) == (p_al is None) and (b_un is None) == (p_un is None)
            if p_al is not None:
                exp = ref_eng.generate(p_al)
                assert b_al.records().tobytes() == exp.records().tobytes() and b_al.errlog().tobytes() == exp.errlog().tobytes()
                assert np.array_equal(b_al.reads(), exp.reads()) and int(b_al.info.n_reads) == n_al
            if p_un is not None:
                exp = ref_eng.generate(p_un)
                assert b_un.eng is eng.step_engine() and b_un.records().tobytes() == exp.records().tobytes()
                assert np.array_equal(b_un.reads(), exp.reads()) and int(b_un.info.record_bytes) == int(exp.info.record_bytes) > 0
        # wrong halves, and the companion's tables belong to its owner
        with pytest.raises(E.EngineError):
            eng.generate_step(E.make_params(seed=5, first_read=0, n_reads=10, kind=E.NS_KIND_UNALIGNED, max_len=small_ref.max_chrom), None)
        with pytest.raises(E.EngineError) as ei:
            eng.step_engine().load_model(small_model)
        assert ei.value.code == E.NS_ESTATE
        # a new model on the owner reaches the companion with the next step
        eng.load_model(small_model)
        b_al, b_un = eng.generate_step(E.make_params(seed=6, first_read=0, n_reads=100, max_len=small_ref.max_chrom),
                                       E.make_params(seed=6, first_read=100, n_reads=100, kind=E.NS_KIND_UNALIGNED, max_len=small_ref.max_chrom))
        assert b_un.records().tobytes() == ref_eng.generate(E.make_params(seed=6, first_read=100, n_reads=100, kind=E.NS_KIND_UNALIGNED, max_len=small_ref.max_chrom)).records().tobytes()
    finally:
        eng.close(); ref_eng.close()


def _bench_line(args, env_extra=None, timeout=900):
    import json
    import subprocess
    import sys
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    env.pop("WORLD_SIZE", None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.split("\n") if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_gpus_2_launches_its_own_ranks_on_one_gpu():
    """`bench.py --gpus N` WITHOUT torchrun starts N ranks itself; here 2 ranks on the one GPU of the box over gloo"""
    d = _bench_line(["--gpus", "2", "--dist-backend", "gloo", "--steps", "2", "--warmup", "1", "--reads", "100000"], {"NS_BENCH_DEVICE": "0"})
    assert d["n_gpus"] == 2 and d["multi_gpu"]["world_size"] == 2 and d["multi_gpu"]["collectives_in_timed_region"] == 0
    assert len(d["multi_gpu"]["per_rank"]) == 2 and d["multi_gpu"]["reference_broadcast_ms"] > 0
    assert d["value"] > 0 and d["scaling"] == "weak" and d["config"]["reads_per_step_per_gpu"] == 100000
    # what a scaling run needs without interpretation: rank 0 alone with the same binary, and the ratio (two ranks on ONE GPU: about a half)
    mg = d["multi_gpu"]
    assert mg["backend_reported"] == "gloo" and mg["single_rank_reference"]["value"] > 0
    assert abs(mg["scaling_efficiency"] - d["value"] / (2 * mg["single_rank_reference"]["value"])) < 1e-9 and 0.2 < mg["scaling_efficiency"] < 1.3


def test_bench_metagenome_and_extras_objects():
    """configs[4] as a bench workload, and the `serial` / `errlog_on` objects of the default line (here at a reduced size)"""
    d = _bench_line(["--metagenome", "--steps", "2", "--warmup", "1", "--reads", "200000", "--no-cpu-baseline", "--no-e2e"])
    assert "zymo10_like" in d["config"]["workload"] and d["value"] > 0 and d["n_gpus"] == 1
    assert d["serial"]["value"] > 0 and d["errlog_on"]["value"] > 0 and d["errlog_on"]["k_errlog_ms"] > 0
    assert 0 < d["errlog_on"]["k_errlog_frac"] < 1 and d["errlog_on"]["errlog_bytes_per_read"] > 1000
    r = d["roofline"]
    assert r["frac_kernel_only_bytes"] < r["frac"] < 1
