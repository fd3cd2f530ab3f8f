#!/usr/bin/env python
"""Drop-in for the simulation stage of NanoSim: same sub-commands, flags, model files and output files as
``src/simulator.py`` of bcgsc/NanoSim v3.2.2 (CLI: S:2070-2218, genome driver: S:2226-2320), with the per-read
loop running on MI355X through the C-ABI (include/nanosim_amd.h).

    python -m nanosim_amd.simulator genome -rg ref.fa -c model/training -o out/simulated -n 100000 [--fastq] ...

Differences a user can see: ``--seed`` is honoured (the reference re-seeds from the OS in
``simulation()``, S:1591-1592, so its --seed has no effect); read numbers are the read's index (no gaps);
``-t K`` keeps the reference's meaning for the OUTPUT (K sub-files written side by side, then concatenated, S:1588-1639) while GPUs,
not processes, do the work (run under ``torch.distributed.run`` for several GPUs).
"""
from __future__ import annotations

import argparse
import os
import sys
import time
from textwrap import dedent
from time import strftime

import numpy as np

from . import engine as E
from . import model as M
from . import shard

VERSION = "3.2.2"
BATCH_READS = 1_000_000
ERR_HEADER = b"Seq_name\tSeq_pos\terror_type\terror_length\tref_base\tseq_base\n"      # S:1634


def log(msg):
    sys.stdout.write(strftime("%Y-%m-%d %H:%M:%S") + ": " + msg + "\n")
    sys.stdout.flush()


MERGE_HELP = ("Not in the reference: with several ranks (one process per GPU) concatenate every rank's sub-files into ONE file per output, "
              "as the reference's workers do (S:1626-1639).  Off by default when WORLD_SIZE > 1: rank 0 would append terabytes through one "
              "inode at 6-10 GB/s after minutes of GPU work; <file>.subfiles lists the parts in order (cat $(cat <file>.subfiles) = <file>)")
NO_MERGE_HELP = ("Not in the reference: with -t K > 1 (or several ranks) keep the sub-files the workers wrote side by side and list them in "
                 "<file>.subfiles (cat $(cat <file>.subfiles) = <file>) instead of concatenating them into one file as the reference does "
                 "(S:1626-1639).  Writes into ONE inode serialise in the kernel (5.7 GB/s on the measured box): the merge, not the GPU, sets "
                 "the wall clock of a large run.  NS_KEEP_SUBFILES=1 does the same")


def build_parser():
    parser = argparse.ArgumentParser(
        description=dedent('''
        Simulation step
        -----------------------------------------------------------
        Given error profiles, reference genome, metagenome,
        and/or transcriptome, simulate ONT DNA or RNA reads
        '''), formatter_class=argparse.RawDescriptionHelpFormatter)
    parser.add_argument('-v', '--version', action='version', version='NanoSim ' + VERSION + ' (nanosim_amd, MI355X)')
    sub = parser.add_subparsers(help="You may run the simulator on genome, transcriptome, or metagenome mode.", dest='mode')

    g = sub.add_parser('genome', help="Run the simulator on genome mode")
    g.add_argument('-rg', '--ref_g', help='Input reference genome', required=True)
    g.add_argument('-c', '--model_prefix', default="training",
                   help='Location and prefix of error profiles generated from characterization step (Default = training)')
    g.add_argument('-o', '--output', default="simulated", help='Output location and prefix for simulated reads (Default = simulated)')
    g.add_argument('-n', '--number', type=int, default=20000, help='Number of reads to be simulated (Default = 20000)')
    g.add_argument('-x', '--coverage', type=float, default=None,
                   help='Coverage of the simulated reads, Note: Coverage will override the number of reads')
    g.add_argument('-max', '--max_len', type=int, default=float("inf"), help='The maximum length for simulated reads (Default = Infinity)')
    g.add_argument('-min', '--min_len', type=int, default=50, help='The minimum length for simulated reads (Default = 50)')
    g.add_argument('-med', '--median_len', type=int, default=None, help='The median read length (Default = None)')
    g.add_argument('-sd', '--sd_len', type=float, default=None, help='The standard deviation of read length in log scale (Default = None)')
    g.add_argument('--seed', type=int, default=None, help='Manually seeds the pseudo-random number generator')
    g.add_argument('-hp', '--homopolymer', action='store_true', default=False, help='Simulate homopolymer lengths (Default = False)')
    g.add_argument('-k', '--KmerBias', type=int, default=None,
                   help='Minimum homopolymer length to simulate homopolymer contraction and expansion events in, a typical k is 5')
    g.add_argument('-s', '--strandness', type=float, default=None,
                   help='Proportion of sense sequences. Overrides the value profiled in characterization stage.')
    g.add_argument('-dna_type', choices=["linear", "circular"], default="linear", help='Specify the dna type: circular OR linear (Default = linear)')
    g.add_argument('--perfect', action='store_true', default=False, help='Ignore error profiles and simulate perfect reads')
    g.add_argument('--fastq', action='store_true', default=False, help='Output fastq files instead of fasta files')
    g.add_argument('--chimeric', action='store_true', default=False, help='Simulate chimeric reads')
    g.add_argument('-t', '--num_threads', type=int, default=1, help='Number of threads for simulation (Default = 1)')
    g.add_argument('--no-merge', dest='no_merge', action='store_true', default=False, help=NO_MERGE_HELP)
    g.add_argument('--merge', dest='merge', action='store_true', default=False, help=MERGE_HELP)

    t = sub.add_parser('transcriptome', help="Run the simulator on transcriptome mode")
    t.add_argument('-rt', '--ref_t', required=True)
    t.add_argument('-rg', '--ref_g', default='')
    t.add_argument('-e', '--exp', required=True)
    t.add_argument('-c', '--model_prefix', default="training")
    t.add_argument('-o', '--output', default="simulated")
    t.add_argument('-n', '--number', type=int, default=20000)
    t.add_argument('-x', '--coverage', type=float, default=None)
    t.add_argument('-max', '--max_len', type=int, default=float("inf"))
    t.add_argument('-min', '--min_len', type=int, default=50)
    t.add_argument('--seed', type=int, default=None)
    t.add_argument('-hp', '--homopolymer', action='store_true', default=False)
    t.add_argument('-k', '--KmerBias', type=int, default=None)
    t.add_argument('-b', '--basecaller', choices=["albacore", "guppy"], default=None)
    t.add_argument('-s', '--strandness', type=float, default=None)
    t.add_argument('--no_model_ir', action='store_false', default=True)
    t.add_argument('--perfect', action='store_true', default=False)
    t.add_argument('--polya', default=None)
    t.add_argument('--fastq', action='store_true', default=False)
    t.add_argument('-t', '--num_threads', type=int, default=1)
    t.add_argument('--no-merge', dest='no_merge', action='store_true', default=False, help=NO_MERGE_HELP)
    t.add_argument('--merge', dest='merge', action='store_true', default=False, help=MERGE_HELP)
    t.add_argument('--uracil', action='store_true', default=False)

    mg = sub.add_parser('metagenome', help="Run the simulator on metagenome mode")
    mg.add_argument('-gl', '--genome_list', required=True)
    mg.add_argument('-a', '--abun', required=True)
    mg.add_argument('-dl', '--dna_type_list', required=True)
    mg.add_argument('-c', '--model_prefix', default="training")
    mg.add_argument('-o', '--output', default="simulated")
    mg.add_argument('-max', '--max_len', type=int, default=float("inf"))
    mg.add_argument('-min', '--min_len', type=int, default=50)
    mg.add_argument('-med', '--median_len', type=int, default=None)
    mg.add_argument('-sd', '--sd_len', type=float, default=None)
    mg.add_argument('--seed', type=int, default=None)
    mg.add_argument('-hp', '--homopolymer', action='store_true', default=False)
    mg.add_argument('-k', '--KmerBias', type=int, default=None)
    mg.add_argument('-s', '--strandness', type=float, default=None)
    mg.add_argument('--perfect', action='store_true', default=False)
    mg.add_argument('--abun_var', nargs='+', type=float, default=None)
    mg.add_argument('--fastq', action='store_true', default=False)
    mg.add_argument('--chimeric', action='store_true', default=False)
    mg.add_argument('-t', '--num_threads', type=int, default=1)
    mg.add_argument('--no-merge', dest='no_merge', action='store_true', default=False, help=NO_MERGE_HELP)
    mg.add_argument('--merge', dest='merge', action='store_true', default=False, help=MERGE_HELP)
    return parser, g, mg, t


def calculate_read_number_from_coverage(ref: M.Reference, model_prefix: str, coverage: float) -> int:
    """S:2024-2068.  The reference estimates the mean read length with 10^7 KDE samples; a Gaussian KDE sample has
    the mean of its training vector, so the expectation is taken directly."""
    rate = None
    with open(model_prefix + "_reads_alignment_rate") as f:
        rate = float(f.readline().strip().split('\t')[1])
    npz = np.load(model_prefix + "_kde.npz") if os.path.exists(model_prefix + "_kde.npz") else None
    al = M._load_kde(model_prefix, "aligned_reads", npz)[0]
    un = M._load_kde(model_prefix, "unaligned_length", npz)[0]
    n_est = 10000000
    n_al = int(n_est * rate / (rate + 1))
    mean = (n_al * float(al.mean()) + (n_est - n_al) * float(un.mean())) / n_est
    return int(ref.genome_len / mean * coverage)


def validate_genome_args(a, parser_g):
    """S:2251-2280: same messages, usage on stderr, exit code 1."""
    def die(msg, to_err=True):
        (sys.stderr if to_err else sys.stdout).write("\n" + msg + "\n")
        parser_g.print_help(sys.stderr)
        sys.exit(1)
    if a.homopolymer and (a.KmerBias is None or a.KmerBias < 0):
        die("Please input proper kmer bias value >= 0 to simulate homopolymer contraction and expansion events from", False)
    if a.strandness and (a.strandness < 0 or a.strandness > 1):
        die("Please input proper strandness value between 0 and 1", False)
    if (a.median_len and not a.sd_len) or (a.sd_len and not a.median_len):
        die("Please provide both mean and standard deviation of read length!")
    if a.median_len and a.sd_len and a.chimeric:
        die("Lognormal distributed reads cannot be chimeric!")
    if a.max_len < a.min_len:
        die("Maximum read length must be longer than Minimum read length!")
    if a.perfect and a.chimeric:
        die("Perfect reads cannot be chimeric", False)


def _cap_stripes(stripes: int, n_outputs: int) -> int:
    """-t K opens 2 x K descriptors per output (two batches in flight): K is capped at 64 and by RLIMIT_NOFILE"""
    k = max(1, min(int(stripes), 64))
    try:
        import resource
        soft = resource.getrlimit(resource.RLIMIT_NOFILE)[0]
        if soft != resource.RLIM_INFINITY:
            k = max(1, min(k, (int(soft) - 64) // max(1, 4 * n_outputs)))
    except (ImportError, ValueError, OSError):
        pass
    return k


def _step_batches(n_al: int, n_un: int) -> tuple[int, int]:
    """reads per worker call of the two phases: one STEP of BATCH_READS reads = an aligned call on its share and an unaligned call on
    the rest (950 000 + 50 000 at the usual 19:1), so that the two contexts of _run_phases work through the run in lockstep — the step
    bench.py times — instead of the unaligned phase racing ahead in one big batch"""
    tot = max(1, n_al + n_un)
    al = max(1000, min(BATCH_READS, int(round(BATCH_READS * n_al / tot))))
    return al, max(1000, BATCH_READS - al)


def _serial_schedule() -> bool:
    """NS_SERIAL=1: aligned and unaligned worker calls one after the other on ONE engine context (A/B runs, GPUs short of memory)"""
    return os.environ.get("NS_SERIAL", "0") != "0"


def _background_engine(device, setup, owner=None):
    """Where the unaligned worker calls of this GPU run, NEXT TO the aligned ones: by default the step companion of `owner`
    (Engine.step_engine: it shares the owner's reference, model and mode tables, nothing is uploaded twice; the library runs it from
    its own worker thread, ns_generate_step); NS_TWO_ENGINES=1: a second engine context of its own (the schedule of rounds 2-4: `setup`
    installs reference, mode tables and model on it).  Returns (engine of the unaligned calls, owner of a StepPair or None)."""
    if owner is not None and os.environ.get("NS_TWO_ENGINES", "0") == "0":
        return owner.step_engine(), owner
    eng = E.Engine(device)
    try:
        eng.set_background(True)
        setup(eng)
    except BaseException:
        eng.close()
        raise
    return eng, None


class StepPair:
    """Pairs the worker calls of the two phases of a run into steps: when the aligned phase asks for its next batch and the unaligned
    phase is waiting with one, both go into ONE ns_generate_step call (include/nanosim_amd.h) — the aligned call on the engine, the
    unaligned one on its step companion, side by side on the GPU; a phase whose partner has nothing to run (it is done, or busy
    queueing its files) calls on its own.  The unaligned phase posts its next request only when it has queued the last batch for its
    files, so the companion's buffers are never written under it."""

    def __init__(self, eng):
        import threading
        self.eng, self.un_eng = eng, eng.step_engine()
        self.cv = threading.Condition()
        self.pending = None          # parameters of an unaligned call waiting for its step
        self.result = None           # ("ok", Batch) | ("alone", None) for the waiting unaligned call
        self.al_done = False
        self.steps = self.alone = 0

    def aligned(self, p):
        with self.cv:
            pu, self.pending = self.pending, None
        if pu is None:
            self.alone += 1
            return self.eng.generate(p)
        try:
            b_al, b_un = self.eng.generate_step(p, pu)
        except BaseException as ex:
            with self.cv:            # the unaligned call repeats on its own (and reports its own error, if it was its error)
                self.result = ("alone", None)
                self.cv.notify_all()
            b_al = getattr(ex, "aligned_batch", None)
            if b_al is None:
                raise
            return b_al              # only the unaligned half failed: the aligned batch is good and its phase keeps its batch size
        with self.cv:
            self.result = ("ok", b_un)
            self.cv.notify_all()
        self.steps += 1
        return b_al

    def unaligned(self, p):
        with self.cv:
            if not self.al_done:
                self.pending = p
                while self.result is None and not (self.al_done and self.pending is not None):
                    self.cv.wait()
                if self.result is not None:
                    kind, b = self.result
                    self.result = None
                    if kind == "ok":
                        return b
                else:
                    self.pending = None          # the aligned phase ended without taking it
        return self.un_eng.generate(p)

    def finish_aligned(self):
        with self.cv:
            self.al_done = True
            self.cv.notify_all()


def _run_phases(aligned, unaligned, rank, n_done, step_owner=None):
    """simulation() (S:1588-1672) starts the unaligned workers once the aligned ones are joined.  That order constrains the FILES only —
    the two phases write different files and a read is a function of (seed, read index) — so here `unaligned` runs in a second host
    thread NEXT TO `aligned`: one GPU, the schedule bench.py times.  step_owner (an Engine): the two phases take their batches through a
    StepPair of it — `aligned(gen)` / `unaligned(gen)` are then called with the pair's generate functions (one ns_generate_step per step);
    None: they are called without arguments (closures over two engine contexts of their own).
    unaligned is None: nothing to run (--perfect).  The log keeps the reference's order."""
    import threading
    err = []
    pair = StepPair(step_owner) if (step_owner is not None and unaligned is not None) else None

    def bg():
        try:
            unaligned(pair.unaligned) if pair else unaligned()
        except BaseException as ex:                      # (SystemExit of a failed merge included: threading would swallow it)
            err.append(ex)
    t = None
    if unaligned is not None:
        t = threading.Thread(target=bg, name="ns-unaligned")
        t.start()
    try:
        aligned(pair.aligned) if pair else aligned()
    finally:
        if pair:
            pair.finish_aligned()
        if t is not None:
            if rank == 0:
                log("Start simulation of random reads")
            t.join()
    if err:
        raise err[0]
    if pair and os.environ.get("NS_CLI_TRACE") is not None:
        sys.stderr.write("[cli] %d steps through ns_generate_step, %d aligned worker calls on their own\n" % (pair.steps, pair.alone))
    if t is not None and rank == 0:
        sys.stdout.write(strftime("%Y-%m-%d %H:%M:%S") + ": Number of reads simulated >> " + str(n_done) + "\n")
        sys.stdout.flush()


def _batch_params(n, first, *, seed, kind, fastq, chimeric, min_len, max_len, median_len, sd_len, want_errlog, kmer_bias, meta, trx, uracil,
                  model_ir, emit_records=True):
    return E.make_params(seed=seed, first_read=first, n_reads=n, kind=kind, fastq=fastq, chimeric=chimeric, kmer_bias=kmer_bias, min_len=min_len,
                         max_len=max_len, median_len=median_len, sd_len=sd_len, emit_records=emit_records, emit_errlog=want_errlog, meta=meta,
                         trx=trx, uracil=uracil, model_ir=model_ir)


def _write_batches(eng, out_path, err_path, *, seed, first, count, kind, fastq, chimeric, min_len, max_len, median_len,
                   sd_len, want_errlog, kmer_bias=0, meta=False, err_header=b"", trx=False, uracil=False, model_ir=False, dist=None, stripes=1,
                   quiet=False, tag="", batch_reads=None, gen=None):
    """Reads [first, first + count) of this rank into out_path (and their error-profile rows into err_path), through the engine's output
    sinks (include/nanosim_amd.h: ns_sink_*): the images of batch i leave the GPU and reach the files while batch i + 1 is generated.

    stripes == 1: rank 0 writes the final files front to back, every other rank one sub-file per output.
    stripes > 1 (-t K; the reference's workers write K sub-files that are concatenated afterwards, S:1588-1639): every batch is cut at read
    boundaries into K sub-files `<out>_aligned_reads<i>.fasta` / `<out>_error_profile<i>` / `<out>_unaligned_reads<i>.fasta` that are written in
    parallel — writes into ONE file serialise on its inode, K files do not — and appended to the final file at the end, in order.
    NS_KEEP_SUBFILES=1 skips that merge: the sub-files stay and `<file>.subfiles` lists them in order (cat $(cat x.subfiles) = x).
    Several ranks: a rank that is done publishes the list of its sub-files (shard.publish_parts); rank 0 appends them in rank order as they
    appear (shard.collect_parts) — no collective, and a rank that fails leaves a marker instead of a hanging peer.
    quiet: no progress line on stdout (the worker call that runs on the background context, _run_phases).
    gen: what generates a batch from its parameters (StepPair.aligned / .unaligned); None: eng.generate."""
    rank = dist.get_rank() if dist is not None else 0
    stripes = _cap_stripes(stripes, len([p for p in (out_path, err_path) if p]))
    world = dist.get_world_size() if dist is not None else 1
    trace = os.environ.get("NS_CLI_TRACE") is not None       # per-batch host timing on stderr
    keep = os.environ.get("NS_KEEP_SUBFILES", "0") != "0"      # (--no-merge sets it: main)
    kw = dict(seed=seed, kind=kind, fastq=fastq, chimeric=chimeric, min_len=min_len, max_len=max_len, median_len=median_len, sd_len=sd_len,
              want_errlog=err_path is not None, kmer_bias=kmer_bias, meta=meta, trx=trx, uracil=uracil, model_ir=model_ir)
    batch = min(getattr(eng, "_batch_reads", BATCH_READS), batch_reads or BATCH_READS)
    paths = [p for p in (out_path, err_path) if p]
    which = {out_path: E.NS_BUF_RECORDS, err_path: E.NS_BUF_ERRLOG}
    files = {p: [] for p in paths}                           # what holds this rank's bytes of p, in order
    open_now, prev = [], []                                  # (sink, fd) of the batch being queued / of the batch before it
    single = {}                                              # stripes == 1: the one sink per output
    done = n_sub = 0

    def close_all(lst):
        """every (sink, fd) of the list is closed exactly once, whatever fails on the way; the first error is raised at the end"""
        first_err = None
        while lst:
            sk, fd = lst.pop(0)
            try:
                sk.close()                               # waits for the file writes; raises on ENOSPC & co
            except Exception as ex:
                first_err = first_err or ex
            finally:
                try:
                    os.close(fd)
                except OSError as ex:
                    first_err = first_err or ex
        if first_err is not None:
            raise first_err
    try:
        if stripes == 1:
            for p in paths:
                name = shard.part_path(p, rank)
                fd = os.open(name, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
                if os.environ.get("NS_CLI_DROP_OUTPUT", "0") != "0":       # measurement aid: the bytes cross PCIe and are dropped (empty files)
                    os.close(fd)
                    fd = os.open("/dev/null", os.O_WRONLY)
                single[p] = eng.sink(fd)
                open_now.append((single[p], fd))
                files[p].append(name)
            if err_path and err_header and rank == 0:   # rank 0 opens the error profile with the column header (S:1634)
                single[err_path].put(err_header)
        while done < count:
            n = min(batch, count - done)
            t0 = time.perf_counter()
            try:
                b = (gen or eng.generate)(_batch_params(n, first + done, **kw))
            except E.EngineError as ex:              # not enough free HBM for this batch size (shared GPU): halve it and go on
                if getattr(ex, "code", 0) != E.NS_ENOMEM or n <= 1000 or meta:
                    raise                            # (metagenome workers: the batches are part of the result)
                batch = eng._batch_reads = max(1000, n // 2)
                continue
            t1 = time.perf_counter()
            if stripes == 1:
                for p in paths:
                    single[p].write(which[p])
            else:
                cuts = sorted({k * n // stripes for k in range(stripes + 1)})
                offs = dict(zip(paths, b.record_offsets(cuts)))
                for lo, hi in zip(range(len(cuts) - 1), range(1, len(cuts))):
                    for p in paths:
                        name = shard.subfile_path(p, n_sub if world == 1 else "%d_%d" % (rank, n_sub))
                        fd = os.open(name, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
                        sk = eng.sink(fd)
                        open_now.append((sk, fd))
                        files[p].append(name)
                        if p == err_path and err_header and rank == 0 and n_sub == 0:
                            sk.put(err_header)
                        sk.write(which[p], int(offs[p][lo]), int(offs[p][hi] - offs[p][lo]))
                    n_sub += 1
                close_all(prev)                      # the batch before this one is in its files; this one's copies are under way
                prev, open_now = open_now, []
            if trace:
                c = eng.io_counters()
                # device: first to last event of the worker call — it includes what the HOST does in between (the first-use hipMalloc of a result
                # slot, the wait for a slot whose last batch is still crossing PCIe); kernels: the sum of the kernel phases alone
                sys.stderr.write("[cli%s] batch %d reads: generate %.1f ms (device %.1f, kernels %.1f), %.2f GB queued; so far %.2f GB copied at %s GB/s (DMA), copier waited "
                                 "%.2f s for staging, writers %.2f s in pwrite\n"
                                 % (tag, n, (t1 - t0) * 1e3, b.info.ms_total, sum(float(x) for x in b.info.ms_kernel), (int(b.info.record_bytes) + int(b.info.errlog_bytes)) / 1e9, c["bytes"] / 1e9,
                                    "%.1f" % c["d2h_gbs"] if c["d2h_gbs"] else "-", c["wait_staging_s"], c["write_s"]))
            done += n
            if rank == 0 and not quiet:
                sys.stdout.write(strftime("%Y-%m-%d %H:%M:%S") + ": Number of reads simulated >> " + str(first + done) + "\r")
                sys.stdout.flush()
        close_all(prev)
        close_all(open_now)
        if stripes > 1 and err_path and err_header and rank == 0 and not files[err_path]:     # no reads at all: the header alone
            name = shard.subfile_path(err_path, 0 if world == 1 else "0_0")
            with open(name, "wb") as f:
                f.write(err_header)
            files[err_path].append(name)
        if rank:
            for p in paths:
                shard.publish_parts(p, rank, files[p])
    except BaseException as ex:
        for lst in (prev, open_now):
            try:
                close_all(lst)
            except Exception:
                pass
        if world > 1:
            for p in paths:
                shard.mark_failed(p, rank, repr(ex))
        raise
    if rank == 0:
        if not quiet:
            sys.stdout.write('\n')
        for p in paths:
            shard.collect_parts(p, world, files[p], keep=keep)


def run_genome(a, parser_g):
    validate_genome_args(a, parser_g)
    rank, device, world, dist, bdev = shard.init_dist()
    if a.KmerBias and not a.homopolymer:
        sys.stderr.write("\n-k/--KmerBias needs -hp (the reference crashes on the missing homopolymer parameters, S:504,639)\n")
        sys.exit(1)
    if rank == 0:
        print("\nrunning the code with following parameters:\n")
        for k in ("ref_g", "model_prefix"):
            print(k, getattr(a, k))
        print("out", a.output); print("number", [a.number]); print("coverage", a.coverage); print("perfect", a.perfect)
        print("homopolymer", a.homopolymer); print("dna_type", a.dna_type); print("strandness", a.strandness)
        print("sd_len", a.sd_len); print("median_len", a.median_len); print("max_len", a.max_len); print("min_len", a.min_len)
        print("fastq", a.fastq); print("chimeric", a.chimeric); print("num_threads", max(a.num_threads, 1))
        log(' '.join(sys.argv))
    out = a.output
    d = os.path.dirname(out)
    if d:
        os.makedirs(d, exist_ok=True)
    if rank == 0:
        log("Read in reference ")
    eng = E.Engine(device)
    ref = M.read_fasta(a.ref_g, a.dna_type) if rank == 0 else None
    ext = ".fastq" if a.fastq else ".fasta"
    seed = a.seed if a.seed is not None else int.from_bytes(os.urandom(8), "little") >> 1
    # S:354-356; every rank leaves together (the verdict of rank 0's check travels in the header of the broadcast)
    bad = "Do not choose circular if there is more than one chromosome in the genome!\n" if (rank == 0 and len(ref.names) > 1 and a.dna_type == "circular") else None
    keep = None
    outputs = [out + "_aligned_reads" + ext, out + "_aligned_error_profile", out + "_unaligned_reads" + ext]
    if dist is not None:
        shard.clean_parts(outputs, rank)
        ref, keep, extra = shard.broadcast_reference(ref, dist, device=bdev, extra=dict(seed=seed), error=bad)
        seed = extra["seed"]               # rank 0's: a read is a function of (seed, read index)
        if bdev is None:                                                                    # (gloo: the bases arrived in host memory)
            ref = M.Reference(ref.names, keep.numpy(), ref.chrom_off, ref.circular)
    else:
        shard.agree(None, bad is None, bad or "")
    with shard.failure_markers(outputs, rank, world):
        if rank == 0:
            log("Read error profile" if not a.perfect else "Read KDF of aligned reads")
        mdl = M.load_model(a.model_prefix, perfect=a.perfect, strandness=a.strandness, chimeric=a.chimeric,
                           homopolymer=a.homopolymer, fastq=a.fastq)

        def setup(e):
            if dist is not None and bdev is not None:
                e.set_reference_device(keep.data_ptr(), ref)
            else:
                e.set_reference(ref)
            e.load_model(mdl)
        setup(eng)
        number = a.number
        if a.coverage is not None:
            if rank == 0:
                print("\nCalculating the number of reads to be simulated based on the coverage, if you specified the number of reads "
                      "concurrently with the coverage, coverage will override number of reads.\n")
            number = calculate_read_number_from_coverage(ref, a.model_prefix, a.coverage)
        n_al, n_un = mdl.split_counts(number)
        max_len = int(min(a.max_len, ref.max_chrom))                                            # S:2318
        kind = E.NS_KIND_PERFECT if a.perfect else E.NS_KIND_ALIGNED
        if rank == 0:
            if a.median_len and a.sd_len:
                log("Simulating read length with log-normal distribution")
            log("Start simulation of aligned reads")
        lo, hi = shard.partition(n_al, world)[rank]
        ulo, uhi = shard.partition(n_un, world)[rank]
        stripes = max(a.num_threads, 1)
        eng_un, step_owner = eng, None
        if not a.perfect and not _serial_schedule():
            eng_un, step_owner = _background_engine(device, setup, eng)
        b_al, b_un = _step_batches(n_al, n_un) if eng_un is not eng else (None, None)

        def aligned(gen=None):
            _write_batches(eng, outputs[0], outputs[1], seed=seed, first=lo, count=hi - lo, kind=kind,
                           fastq=a.fastq, chimeric=a.chimeric, min_len=a.min_len, max_len=max_len, median_len=a.median_len, sd_len=a.sd_len,
                           want_errlog=True, kmer_bias=a.KmerBias or 0, err_header=ERR_HEADER, dist=dist, stripes=stripes, tag=" aligned",
                           batch_reads=b_al, gen=gen)

        def unaligned(quiet, gen=None):                                                         # S:1642-1672
            _write_batches(eng_un, outputs[2], None, seed=seed, first=n_al + ulo, count=uhi - ulo, kind=E.NS_KIND_UNALIGNED,
                           fastq=a.fastq, chimeric=False, min_len=a.min_len, max_len=max_len, median_len=a.median_len, sd_len=a.sd_len,
                           want_errlog=False, dist=dist, stripes=stripes, quiet=quiet, tag=" unaligned", batch_reads=b_un, gen=gen)
        try:
            if a.perfect:
                aligned()
            elif eng_un is eng:
                aligned()
                if rank == 0:
                    log("Start simulation of random reads")
                unaligned(False)
            else:
                _run_phases(aligned, lambda gen=None: unaligned(True, gen), rank, n_al + uhi, step_owner)
        finally:
            if eng_un is not eng:
                eng_un.close()
            eng.close()
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        log("Finished!")


def run_metagenome(a, parser_mg):
    """The metagenome branch of main() (S:2416-2527) + simulation("metagenome") (S:1568-1672).  One ns_generate call plays
    one worker of the reference: it keeps its own per-species base quota (S:835) and numbers its reads consecutively."""
    from . import metagenome as MG
    validate_genome_args(a, parser_mg)
    if a.KmerBias and not a.homopolymer:
        sys.stderr.write("\n-k/--KmerBias needs -hp (the reference crashes on the missing homopolymer parameters, S:504,639)\n")
        sys.exit(1)
    rank, device, world, dist, bdev = shard.init_dist()
    if rank == 0:
        print("\nrunning the code with following parameters:\n")
        for k in ("genome_list", "abun", "dna_type_list", "model_prefix"):
            print(k, getattr(a, k))
        print("out", a.output); print("perfect", a.perfect); print("strandness", a.strandness); print("sd_len", a.sd_len)
        print("median_len", a.median_len); print("max_len", a.max_len); print("min_len", a.min_len); print("abun_var", a.abun_var)
        print("fastq", a.fastq); print("chimeric", a.chimeric); print("num_threads", max(a.num_threads, 1))
        log(' '.join(sys.argv))
    out = a.output
    d = os.path.dirname(out)
    if d:
        os.makedirs(d, exist_ok=True)
    eng = E.Engine(device)
    keep = None
    if rank == 0:
        log("Read in reference ")
        mref = MG.read_metagenome(a.genome_list, a.dna_type_list)
        numbers, samples = MG.read_abundance(a.abun, mref.species)
    seed = a.seed if a.seed is not None else int.from_bytes(os.urandom(8), "little") >> 1
    ext = ".fastq" if a.fastq else ".fasta"
    if dist is not None:
        info = dict(species=mref.species, off=mref.species_chrom_off.tolist(), keys=mref.chrom_names, numbers=numbers,
                    samples=samples, seed=seed) if rank == 0 else None
        # (sub-files of every sample this rank may write: the sample count is only known after the broadcast, so by pattern —
        # ".part<rank>" followed by the end of the name or a dot, so that rank 1 does not take rank 10's files)
        import glob
        import re
        tail = re.compile(r"\.part%d(\.|$)" % rank)
        for q in glob.glob(glob.escape(a.output) + "_sample*.part%d*" % rank) if rank else ():
            if tail.search(q):
                os.unlink(q)
        ref, keep, info = shard.broadcast_reference(mref.ref if rank == 0 else None, dist, device=bdev, extra=info)
        if bdev is None:
            ref = M.Reference(ref.names, keep.numpy(), ref.chrom_off, ref.circular)
        mref = MG.MetaReference(ref, info["species"], np.array(info["off"], dtype=np.uint32), info["keys"])
        numbers, samples, seed = info["numbers"], info["samples"], info["seed"]
    tails = ("_aligned_reads" + ext, "_aligned_error_profile", "_unaligned_reads" + ext)
    outputs = [out + "_sample%d%s" % (s, t) for s in range(len(samples)) for t in tails]
    with shard.failure_markers(outputs, rank, world):
        if rank == 0:
            log("Read error profile" if not a.perfect else "Read KDF of aligned reads")
        mdl = M.load_model(a.model_prefix, perfect=a.perfect, strandness=a.strandness, chimeric=a.chimeric, fastq=a.fastq,
                           homopolymer=a.homopolymer)

        def setup(e):
            e.set_metagenome(mref, dev_ptr=keep.data_ptr() if (dist is not None and bdev is not None) else None)
            e.load_model(mdl)
        setup(eng)
        eng_un, step_owner = eng, None
        if not a.perfect and not _serial_schedule():
            eng_un, step_owner = _background_engine(device, setup, eng)      # (unaligned reads and gaps take any species: no abundances needed, S:1708)
        max_len = a.max_len
        total_len = mref.total_len()
        first = 0
        try:
            for s, abun in enumerate(samples):
                sample = "sample" + str(s)
                if a.abun_var:                                                                  # S:2497-2506: the species of THIS sample
                    sample_len = {sp: total_len[sp] for sp in abun}
                    u = np.random.default_rng([seed & 0xffffffff, seed >> 32, s]).random(len(sample_len))
                    abun = MG.add_abundance_var(abun, sample_len, float(a.abun_var[0]), float(a.abun_var[1]), iter(u.tolist()))
                infl = {sp: MG.inflate_abun(abun, sp, mdl.abun_inflation) for sp in abun} if a.chimeric else None   # S:2510-2514
                eng.set_abundance(mref, abun, infl)
                if rank == 0:
                    log("Simulating sample " + sample)
                    if a.median_len and a.sd_len:
                        log("Simulating read length from log-normal distribution")
                    log("Start simulation of aligned reads")
                n_al, n_un = mdl.split_counts(numbers[s])
                max_len = int(min(max_len, mref.max_chrom))                                     # S:2525
                base = out + "_" + sample
                lo, hi = shard.partition(n_al, world)[rank]
                ulo, uhi = shard.partition(n_un, world)[rank]

                def aligned(gen=None, base=base, first=first, lo=lo, hi=hi, max_len=max_len):
                    _write_batches(eng, base + "_aligned_reads" + ext, base + "_aligned_error_profile", seed=seed, first=first + lo,
                                   count=hi - lo, kind=E.NS_KIND_PERFECT if a.perfect else E.NS_KIND_ALIGNED, fastq=a.fastq, chimeric=a.chimeric,
                                   min_len=a.min_len, max_len=max_len, median_len=a.median_len, sd_len=a.sd_len, want_errlog=True, meta=True,
                                   kmer_bias=0 if a.perfect else (a.KmerBias or 0), err_header=ERR_HEADER, dist=dist, tag=" aligned", gen=gen)

                def unaligned(quiet, gen=None, base=base, first=first, n_al=n_al, ulo=ulo, uhi=uhi, max_len=max_len):     # S:1642
                    _write_batches(eng_un, base + "_unaligned_reads" + ext, None, seed=seed, first=first + n_al + ulo, count=uhi - ulo,
                                   kind=E.NS_KIND_UNALIGNED, fastq=a.fastq, chimeric=False, min_len=a.min_len, max_len=max_len,
                                   median_len=a.median_len, sd_len=a.sd_len, want_errlog=False, meta=True, dist=dist, quiet=quiet, tag=" unaligned",
                                   gen=gen)
                if a.perfect:
                    aligned()
                elif eng_un is eng:
                    aligned()
                    if rank == 0:
                        log("Start simulation of random reads")
                    unaligned(False)
                else:
                    _run_phases(aligned, lambda gen=None, un=unaligned: un(True, gen), rank, first + n_al + uhi, step_owner)
                first += n_al + n_un            # samples draw from disjoint read-index ranges of the same seed
        finally:
            if eng_un is not eng:
                eng_un.close()
            eng.close()
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        log("Finished!")


def run_transcriptome(a, parser_t):
    """The transcriptome branch of main() (S:2322-2414) + simulation("transcriptome") (S:1568-1672)."""
    from . import intron_retention as IR
    from . import transcriptome as TR

    def die(msg, to_err=True):
        (sys.stderr if to_err else sys.stdout).write("\n" + msg + "\n")
        parser_t.print_help(sys.stderr)
        sys.exit(1)
    if a.homopolymer and (a.KmerBias is None or a.KmerBias < 0):                              # S:2350-2354
        die("Please input proper kmer bias value >= 0 to simulate homopolymer contraction and expansion events from", False)
    if a.strandness and (a.strandness < 0 or a.strandness > 1):
        die("Please input proper strandness value between 0 and 1", False)
    if a.max_len < a.min_len:
        die("Maximum read length must be longer than Minimum read length!")
    model_ir = a.no_model_ir                                                                  # store_false: True unless --no_model_ir
    if model_ir and a.ref_g == '':
        die("Please provide a reference genome to simulate intron retention events!")
    if a.polya and a.basecaller is None:
        die("Please input basecaller to simulate polyA tails from.", False)
    if a.KmerBias and not a.homopolymer:
        sys.stderr.write("\n-k/--KmerBias needs -hp (the reference crashes on the missing homopolymer parameters, S:504,639)\n")
        sys.exit(1)
    rank, device, world, dist, bdev = shard.init_dist()
    if rank == 0:
        print("\nrunning the code with following parameters:\n")
        print("ref_g", a.ref_g); print("ref_t", a.ref_t); print("exp", a.exp); print("model_prefix", a.model_prefix); print("out", a.output)
        print("number", [a.number]); print("coverage", a.coverage); print("perfect", a.perfect); print("homopolymer", a.homopolymer)
        print("model_ir", model_ir); print("dna_type", "transcriptome"); print("strandness", a.strandness); print("max_len", a.max_len)
        print("min_len", a.min_len); print("uracil", a.uracil); print("polya", a.polya)
        if a.polya:
            print("basecaller", a.basecaller)
        print("fastq", a.fastq); print("num_threads", max(a.num_threads, 1))
        log(' '.join(sys.argv))
    out = a.output
    d = os.path.dirname(out)
    if d:
        os.makedirs(d, exist_ok=True)
    eng = E.Engine(device)
    keep = None
    if rank == 0:
        log("Read in reference ")
        tr = TR.read_transcriptome(a.ref_t, a.exp, a.polya, a.basecaller)
    seed = a.seed if a.seed is not None else int.from_bytes(os.urandom(8), "little") >> 1
    ext = ".fastq" if a.fastq else ".fasta"
    if dist is not None:
        info = dict(ec=tr.expr_chrom, cum=tr.expr_cum, w=tr.expr_weight, pa=tr.polya, sc=tr.polya_scale, seed=seed) if rank == 0 else None
        shard.clean_parts([out + "_aligned_reads" + ext, out + "_aligned_error_profile", out + "_unaligned_reads" + ext], rank)
        ref, keep, info = shard.broadcast_reference(tr.ref if rank == 0 else None, dist, device=bdev, extra=info)
        if bdev is None:
            ref = M.Reference(ref.names, keep.numpy(), ref.chrom_off, ref.circular)
            keep = None
        tr = TR.TranscriptomeReference(ref, info["ec"], info["cum"], info["w"], info["pa"], info["sc"])
        seed = info["seed"]
    ir = None
    if model_ir:                                                                              # S:403-452
        if rank == 0:
            log("Read in reference genome, IR markov model and GFF3 annotation file")
        ir = IR.load(a.model_prefix, a.ref_g, tr.ref)
        tr = TR.restrict_expression(tr, ir.eligible)                                          # S:1093-1099
    outputs = [out + "_aligned_reads" + ext, out + "_aligned_error_profile", out + "_unaligned_reads" + ext]
    with shard.failure_markers(outputs, rank, world):
        if rank == 0:
            log("Read error profile" if not a.perfect else "Read KDF of aligned reads")
        mdl = M.load_model(a.model_prefix, perfect=a.perfect, strandness=a.strandness, fastq=a.fastq, transcriptome=True,
                           homopolymer=a.homopolymer)

        def setup(e, with_ir=True):
            e.set_transcriptome(tr, dev_ptr=keep.data_ptr() if keep is not None else None)
            if ir is not None and with_ir:
                e.set_intron_retention(ir)
            e.load_model(mdl)
        setup(eng)
        number = a.number
        if a.coverage is not None:
            print("\nCalculating the number of reads to be simulated based on the coverage, if you specified the number of reads "
                  "concurrently with the coverage, coverage will override number of reads.\n")
            number = calculate_read_number_from_coverage(tr.ref, a.model_prefix, a.coverage)
        n_al, n_un = mdl.split_counts(number)
        max_len = int(min(a.max_len, tr.ref.max_chrom))                                           # S:2411
        if rank == 0:
            log("Start simulation of aligned reads")
        lo, hi = shard.partition(n_al, world)[rank]
        ulo, uhi = shard.partition(n_un, world)[rank]
        stripes = max(a.num_threads, 1)
        eng_un, step_owner = eng, None
        if not a.perfect and not _serial_schedule():
            eng_un, step_owner = _background_engine(device, lambda e: setup(e, with_ir=False), eng)     # (S:1156: unaligned reads are never spliced)
        b_al, b_un = _step_batches(n_al, n_un) if eng_un is not eng else (None, None)

        def aligned(gen=None):
            _write_batches(eng, outputs[0], outputs[1], seed=seed, first=lo, count=hi - lo,
                           kind=E.NS_KIND_PERFECT if a.perfect else E.NS_KIND_ALIGNED, fastq=a.fastq, chimeric=False, min_len=a.min_len,
                           max_len=max_len, median_len=None, sd_len=None, want_errlog=True, trx=True, uracil=a.uracil, kmer_bias=a.KmerBias or 0,
                           err_header=ERR_HEADER, model_ir=model_ir, dist=dist, stripes=stripes, tag=" aligned", batch_reads=b_al, gen=gen)

        def unaligned(quiet, gen=None):                                                           # S:1642-1672
            _write_batches(eng_un, outputs[2], None, seed=seed, first=n_al + ulo, count=uhi - ulo,
                           kind=E.NS_KIND_UNALIGNED, fastq=a.fastq, chimeric=False, min_len=a.min_len, max_len=max_len, median_len=None,
                           sd_len=None, want_errlog=False, trx=True, uracil=a.uracil, dist=dist, stripes=stripes, quiet=quiet, tag=" unaligned",
                           batch_reads=b_un, gen=gen)
        try:
            if a.perfect:
                aligned()
            elif eng_un is eng:
                aligned()
                if rank == 0:
                    log("Start simulation of random reads")
                unaligned(False)
            else:
                _run_phases(aligned, lambda gen=None: unaligned(True, gen), rank, n_al + uhi, step_owner)
        finally:
            if eng_un is not eng:
                eng_un.close()
            eng.close()
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        log("Finished!")


def main(argv=None):
    parser, parser_g, parser_mg, parser_t = build_parser()
    if len(sys.argv if argv is None else argv) == (1 if argv is None else 0):
        parser.print_help(sys.stderr)
        sys.exit(1)
    a = parser.parse_args(argv)
    if getattr(a, "no_merge", False) and getattr(a, "merge", False):
        sys.stderr.write("--merge and --no-merge exclude each other\n")
        sys.exit(1)
    world = int(os.environ.get("WORLD_SIZE", "1") or 1)
    if getattr(a, "no_merge", False):
        os.environ["NS_KEEP_SUBFILES"] = "1"
    elif world > 1 and not getattr(a, "merge", False) and "NS_KEEP_SUBFILES" not in os.environ:
        # several ranks: the parts stay where the ranks wrote them unless --merge asks for the reference's single files (MERGE_HELP)
        os.environ["NS_KEEP_SUBFILES"] = "1"
    if a.mode == "genome":
        run_genome(a, parser_g)
    elif a.mode == "metagenome":
        run_metagenome(a, parser_mg)
    elif a.mode == "transcriptome":
        run_transcriptome(a, parser_t)
    else:
        parser.print_help(sys.stderr)
        sys.exit(1)


if __name__ == "__main__":
    main()
