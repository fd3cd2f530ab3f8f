#!/usr/bin/env python
"""bench.py — simulated reads/s (+ bases/s) of the genome-mode hot path on N MI355X.

A "step" = one genome-mode pass over one batch of read indices of BASELINE.json configs[1]: the aligned worker
call (src/simulator.py:1266-1454) on 950 000 reads and the unaligned one (S:1482-1549) on 50 000 (the model's
alignment rate 19:1), E. coli-like 4.64 Mb circular genome, hg002-like error model (mean aligned length ~8.4 kb,
~265 error events/read), FASTA records, reference + model resident in HBM, outputs left in HBM; the two calls run side by side on two engine contexts of the GPU.  N>1: one
process per GPU, read-index ranges sharded, ONE RCCL broadcast of the reference before the timed region, no
collective inside it (weak scaling).

The same JSON line also carries (N = 1):
  "configs2"  BASELINE configs[2] — chr1-size reference, FASTQ, -hp -k 5 — a few steps of the same protocol, with its own roofline;
  "e2e"       the END-TO-END legs of SURVEY section 8(d): generation + device-to-host copy + file writes through the engine's output
              sinks (include/nanosim_amd.h: ns_sink_*), to /dev/null and to files on /dev/shm, with and without the error profile
              the reference always writes, next to the measured page-locked device-to-host rate.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--reads R]
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SEED = 20260926
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PROFILE_ROUNDS = ("r03", "r02")


REFERENCE_PYTHON = {        # the reference itself (bcgsc/NanoSim v3.2.2, simulator.py -t 8), measured in the build container
    "reads_per_s_per_core": {"fasta": 190, "fastq_hp_k5": 44},
    "hardware": "8 vCPU Intel Xeon @ 2.10 GHz (build container, not the GPU box), Python 3.10 / numpy 2.2",
    "note": "the reference's Python cannot travel to the GPU box; its C restatement (oracle/) is timed there instead",
}


def _cpu_worker(args):
    """one host core: its share of the sample through the C restatement of the reference (oracle/ns_oracle.c)"""
    idx, n_al, n_un, fastq, kmer = args
    from tests import oracle_lib
    mdl, ref, eng = _CPU_CTX
    bases = 0
    first = idx * (n_al + n_un)
    for kind, cnt in ((eng.NS_KIND_ALIGNED, n_al), (eng.NS_KIND_UNALIGNED, n_un)):
        done = 0
        while done < cnt:                            # 1 000 reads per call: the buffers of a call stay below ~50 MB per process
            m = min(1000, cnt - done)
            p = eng.make_params(seed=SEED, first_read=first, n_reads=m, kind=kind, max_len=ref.max_chrom, fastq=fastq,
                                kmer_bias=kmer if kind == eng.NS_KIND_ALIGNED else 0)
            per = max(120000 if fastq else 60000, 8_000_000 // m)     # a single FASTQ record of a long read needs more than the average
            bases += int(oracle_lib.generate(mdl, ref, p, bytes_per_read=per)["total_bases"])
            done += m; first += m
    return bases


_CPU_CTX = None


def cpu_baseline(model, ref, engine_mod, per_core, fastq, kmer):
    """The CPU restatement (oracle, kind="port") on ALL host cores — one process per core, as the reference's -t fan-out
    (src/simulator.py:1588-1605) — on a bounded sample of the same workload: per core `per_core` reads in the model's
    aligned : unaligned proportion."""
    import multiprocessing as mp
    global _CPU_CTX
    from tests import oracle_lib
    oracle_lib.lib()                                 # dlopen before the fork
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    n_al, n_un = model.split_counts(per_core)
    _CPU_CTX = (model, ref, engine_mod)
    cpu = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    ctx = mp.get_context("fork")
    with ctx.Pool(cores) as pool:
        pool.map(_cpu_worker, [(i, 20, 0, fastq, kmer) for i in range(cores)])         # start the workers, touch the tables
        t0 = time.perf_counter()
        bases = pool.map(_cpu_worker, [(i, n_al, n_un, fastq, kmer) for i in range(cores)])
        dt = time.perf_counter() - t0
    return dict(value=cores * per_core / dt, unit="reads/s", cores=cores, kind="port", cpu=cpu,
                sample="%d cores x %d reads (%d aligned + %d unaligned each) of the same workload through oracle/ns_oracle.c, one process "
                       "per core, %.1f s" % (cores, per_core, n_al, n_un, dt),
                bases_per_s=sum(bases) / dt, reads_per_s_per_core=per_core / dt, reference_python=REFERENCE_PYTHON)


def measured_traffic(genome, fastq, kmer, kernels):
    """HBM bytes per read of the kernels behind the roofline stage, from the rocprofv3 PMC passes of THIS configuration
    (scripts/profile_round.sh -> profiles/<round>/pmc_<config>.json: FETCH_SIZE x 2 per the gfx950 note + WRITE_SIZE, separate passes).
    PMC counters cannot be collected from inside this process; None when the configuration has not been profiled."""
    key = "%s_%s%s" % (genome, "fastq" if fastq else "fasta", "_k%d" % kmer if kmer else "")
    for rnd in PROFILE_ROUNDS:
        path = os.path.join(ROOT, "profiles", rnd, "pmc_%s.json" % key)
        try:
            pm = json.load(open(path))
        except (OSError, ValueError):
            continue
        tot = 0.0
        for kname, kv in pm.get("kernels", {}).items():
            if kname.startswith(kernels) and "hbm_bytes_per_read" in kv:
                tot += kv["hbm_bytes_per_read"]
        if tot > 0:
            return tot, os.path.relpath(path, ROOT)
    return None, None


class Workload:
    """one configuration resident on this rank's GPU: engines (aligned + background unaligned context), model, reference"""

    def __init__(self, a, genome, fastq, kmer, local_rank, rank, world, dist, serial, aligned_only, tmp):
        import numpy as np
        import torch
        from nanosim_amd import engine, model, synth
        self.engine, self.genome, self.fastq, self.kmer, self.rank, self.world = engine, genome, fastq, kmer, rank, world
        prefix = os.path.join(tmp, "hg002_like")
        if not os.path.exists(prefix + "_kde.npz"):
            synth.write_model(prefix, synth.SynthModelSpec(n_train=1_000_000, seed=SEED), write_pkl=False)
        self.mdl = model.load_model(prefix, fastq=fastq, homopolymer=kmer > 0)
        names = ["ecoli-like"] if genome == "ecoli" else ["chr1-like"]
        self.glen = glen = synth.ECOLI_LEN if genome == "ecoli" else synth.CHR1_LEN
        ref_meta = model.Reference(names, np.zeros(0, np.uint8), np.array([0, glen], dtype=np.uint64),
                                   np.array([1 if genome == "ecoli" else 0], dtype=np.uint8))
        self.eng = engine.Engine(local_rank)
        # the unaligned worker call of a step runs next to the aligned one on its own engine context (own HIP streams and buffers on the
        # same GPU, own host thread) — the way the reference runs its workers side by side (-t, S:1588-1605)
        self.eng_un = None if (aligned_only or serial) else engine.Engine(local_rank)
        if self.eng_un is not None:
            self.eng_un.set_background(True)     # its kernels share the GPU with the aligned call's: few issue slots matter more than a short latency
        self.engs = [e for e in (self.eng, self.eng_un) if e is not None]
        self.broadcast_ms = None
        if world > 1:
            # the reference lives on rank 0; ONE broadcast over xGMI puts it in every GPU's HBM
            bdev = "cuda" if a.dist_backend == "nccl" else "cpu"
            buf = torch.empty(glen, dtype=torch.uint8, device=bdev)
            if rank == 0:
                seq = synth.synth_sequence(glen, SEED, n_frac=0.0005, iupac_frac=0.0002, lower_frac=0.02, hp_boost=0.005)
                buf.copy_(torch.from_numpy(seq))
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            dist.broadcast(buf, src=0)
            torch.cuda.synchronize()
            self.broadcast_ms = (time.perf_counter() - t0) * 1e3
            buf = buf.cuda()
            torch.cuda.synchronize()
            for e in self.engs:
                e.set_reference_device(buf.data_ptr(), ref_meta)
            self.ref_host = None
        else:
            seq = synth.synth_sequence(glen, SEED, n_frac=0.0005, iupac_frac=0.0002, lower_frac=0.02, hp_boost=0.005)
            self.ref_host = model.Reference(names, seq, ref_meta.chrom_off, ref_meta.circular)
            for e in self.engs:
                e.set_reference(self.ref_host)
        for e in self.engs:
            e.load_model(self.mdl)
        self.max_len = min(glen, 1 << 30)
        self.unaligned_delay_s = max(0.0, getattr(a, "unaligned_delay_ms", 0.0)) * 1e-3

    def split(self, n, aligned_only):
        return (n, 0) if aligned_only else self.mdl.split_counts(n)

    def step(self, i, n, n_al, n_un, errlog=False, records=True, after_aligned=None, after_unaligned=None):
        """a step = one genome-mode pass of this GPU over n read indices: the aligned worker call (simulation_aligned_genome,
        S:1266-1454) on round(n r / (r + 1)) reads and the unaligned one (simulation_unaligned, S:1482-1549) on the rest (the model's
        alignment rate r = 19), as simulation() runs them (S:1571-1672).  after_*: called with the batch right after its worker call
        (the end-to-end legs queue the result buffers for their files there)."""
        engine = self.engine
        base = (i * self.world + self.rank) * n
        out = [None, None]

        def aligned():
            b = self.eng.generate(engine.make_params(seed=SEED, first_read=base, n_reads=n_al, fastq=self.fastq, max_len=self.max_len,
                                                     emit_errlog=errlog, kmer_bias=self.kmer, emit_records=records))
            out[0] = b.info
            if after_aligned:
                after_aligned(b)

        def unaligned(e):
            b = e.generate(engine.make_params(seed=SEED, first_read=base + n_al, n_reads=n_un, kind=engine.NS_KIND_UNALIGNED,
                                              fastq=self.fastq, max_len=self.max_len, emit_records=records))
            out[1] = b.info
            if after_unaligned:
                after_unaligned(b)
        if not n_un:
            aligned()
            return out[:1]
        if self.eng_un is None:                                # --serial: one engine, one call after the other
            aligned(); unaligned(self.eng)
            return out
        def late_unaligned():
            if self.unaligned_delay_s > 0:
                time.sleep(self.unaligned_delay_s)
            unaligned(self.eng_un)
        t = threading.Thread(target=late_unaligned)      # (the C call releases the GIL)
        t.start(); aligned(); t.join()
        return out

    def close(self):
        for e in self.engs:
            e.close()


def timed_steps(w, a, n, n_al, n_un, steps, warmup, dist, errlog):
    import torch
    for i in range(warmup):
        w.step(i, n, n_al, n_un, errlog=errlog)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    infos = [w.step(warmup + i, n, n_al, n_un, errlog=errlog) for i in range(steps)]
    torch.cuda.synchronize()
    dt_local = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    return infos, dt, dt_local


def summarise(w, a, infos, dt, n, n_al, n_un, steps, warmup, world, tot_bases, errlog):
    import numpy as np
    engine = w.engine
    al = [st[0] for st in infos]
    # roofline of the dominant kernel (of the aligned worker call): algorithmic bytes per launch / its mean HIP-event duration
    kms = {}
    for k, nm in enumerate(engine.KERNEL_NAMES):
        kms[nm] = float(np.mean([x.ms_kernel[k] for x in al]))
    dom = max(kms, key=kms.get)
    per_launch = np.mean([int(x.total_ref_bases) + int(x.total_bases) * (2 if w.fastq else 1) +
                          16 * int(x.events_used) + 32 * int(x.n_reads) for x in al])
    achieved = per_launch / (kms[dom] * 1e-3) / 1e9
    device_ms = float(np.mean([sum(x.ms_total for x in st) for st in infos]))
    stage = {"k_materialise": ("k_materialise",), "k_hp": ("k_hp", "k_materialise<true, 1>", "k_materialise<false, 1>")}.get(dom, (dom,))
    per_read, traffic_src = measured_traffic(w.genome, w.fastq, w.kmer, stage)
    out = {
        "metric": "simulated reads/sec (genome mode, mean 8 kb)", "value": world * n * steps / dt, "unit": "reads/s",
        "bases_per_s": tot_bases / dt,
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": ("configs[1]: ecoli_like 4,641,652 bp circular" if w.genome == "ecoli" else "configs[2]: chr1_like 248,956,422 bp linear") +
                               ", hg002_like error model, genome mode, %s%s, %d reads/GPU/step = %d aligned + %d unaligned (alignment rate 19:1)"
                               % ("FASTQ" if w.fastq else "FASTA", ", -hp -k %d" % w.kmer if w.kmer else "", n, n_al, n_un),
                   "reads_per_step_per_gpu": n, "aligned_per_step": n_al, "unaligned_per_step": n_un, "errlog": bool(errlog),
                   "errlog_note": "the error-profile text (the reference always writes it, S:2006-2008: ~26 KB per read, 3x the reads) is "
                                  "formatted by k_errlog only when asked for (--errlog; the CLI always asks): it is a file-format stage "
                                  "behind the path the metric names (SURVEY section 8 f-1); the e2e legs below time it",
                   "seed": SEED, "parallelism": "read-index sharding x%d, 1 RCCL broadcast of the reference" % world, "engines_per_gpu": len(w.engs),
                   "engines_note": "aligned and unaligned worker call of a step run side by side on two engine contexts of the GPU, the unaligned one as a background context (ns_set_background); --serial: one after the other on one"},
        "device_ms_per_step": device_ms,
        "aligned_batch": {"reads": n_al, "device_ms": float(np.mean([x.ms_total for x in al])),
                          "reads_per_s_device": n_al / (float(np.mean([x.ms_total for x in al])) * 1e-3), "kernel_ms": kms},
        "kernel_ms": kms,
        "roofline": {"bound": "hbm", "kernel": dom + (" (stage: k_materialise + k_materialise_slow; k_names runs next to it on a second stream)" if dom == "k_materialise" else ""), "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": per_read * n_al if per_read else None,
                     "traffic_source": (traffic_src + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this configuration, per read x reads per launch)") if per_read else None,
                     "algorithmic_bytes_per_launch": float(per_launch),
                     "all_kernels_achieved": per_launch / (sum(kms.values()) * 1e-3) / 1e9},
    }
    if n_un:
        un = [st[1] for st in infos]
        out["unaligned_batch"] = {"reads": n_un, "device_ms": float(np.mean([x.ms_total for x in un])),
                                  "kernel_ms": {nm: float(np.mean([x.ms_kernel[k] for x in un])) for k, nm in enumerate(engine.KERNEL_NAMES)}}
    return out


def e2e_legs(w, n, steps, shm_dir, stripes=16):
    """SURVEY section 8(d) "timing protocol": end to end = generation + device-to-host + file writes.  Every leg: two untimed steps back
    to back (they size both result slots: a 25 GB hipMalloc inside the timed steps cost the first error-profile leg a quarter of its
    rate), then `steps` steps whose record image (and error-profile image) are queued for their files right
    after each worker call (ns_sink_write), then a drain.  reads/s = reads of the timed steps / wall time incl. the drain.
    Destinations: null = /dev/null; shm = ONE file per output on /dev/shm (what the CLI does by default; writes into one inode
    serialise in the kernel); shm<K> = every worker call's images cut at read boundaries into K sub-files per output (the CLI's -t K
    with NS_KEEP_SUBFILES=1: the reference's sub-file layout, S:1588-1639, without the final concatenation)."""
    E = w.engine
    n_al, n_un = w.split(n, False)
    res = {}
    free = None
    try:
        sv = os.statvfs(shm_dir)
        free = sv.f_bavail * sv.f_frsize
    except OSError:
        pass
    for dest in ("null", "shm", "shm%d" % stripes):
        for errlog in (False, True):
            name = "%s_%s%s" % (dest, "fastq" if w.fastq else "fasta", "_errlog" if errlog else "")
            need = (steps + 2) * n * (36_000 if errlog else 9_000) * (2 if w.fastq else 1)
            if dest != "null" and (free is None or free < 3 * need):
                res[name] = {"skipped": "needs %.0f GB on %s" % (need / 1e9, shm_dir)}
                continue
            d = tempfile.mkdtemp(prefix="nsbench_e2e_", dir=shm_dir) if dest != "null" else None
            K = stripes if dest.startswith("shm") and dest != "shm" else 1
            fds, sinks = [], {}
            live = {"al": [], "un": []}            # K > 1: (sink, fd) of the worker call being copied / of the one before it
            counter = [0]
            try:
                def open_sink(eng, fname):
                    fd = os.open("/dev/null" if d is None else os.path.join(d, fname), os.O_WRONLY | (0 if d is None else os.O_CREAT | os.O_TRUNC), 0o644)
                    return eng.sink(fd), fd
                if K == 1:
                    for key, eng, fname in (("al", w.eng, "aligned_reads"), ("err", w.eng, "aligned_error_profile"), ("un", w.eng_un or w.eng, "unaligned_reads")):
                        if key != "err" or errlog:
                            sinks[key], fd = open_sink(eng, fname)
                            fds.append(fd)

                def striped(b, eng, key, with_err):
                    nr = int(b.info.n_reads)
                    cuts = sorted({k * nr // K for k in range(K + 1)})
                    ro, eo = b.record_offsets(cuts)
                    mine = []
                    for lo, hi in zip(range(len(cuts) - 1), range(1, len(cuts))):
                        counter[0] += 1
                        sk, fd = open_sink(eng, "%s_reads%d" % (key, counter[0]))
                        sk.write(E.NS_BUF_RECORDS, int(ro[lo]), int(ro[hi] - ro[lo])); mine.append((sk, fd))
                        if with_err:
                            sk, fd = open_sink(eng, "error_profile%d" % counter[0])
                            sk.write(E.NS_BUF_ERRLOG, int(eo[lo]), int(eo[hi] - eo[lo])); mine.append((sk, fd))
                    for sk, fd in live[key]:             # the worker call before this one is in its files by now (or soon)
                        sk.close(); os.close(fd)
                    live[key] = mine

                def after_al(b):
                    if K > 1:
                        return striped(b, w.eng, "al", errlog)
                    sinks["al"].write(E.NS_BUF_RECORDS)
                    if errlog:
                        sinks["err"].write(E.NS_BUF_ERRLOG)

                def after_un(b):
                    if K > 1:
                        return striped(b, w.eng_un or w.eng, "un", False)
                    sinks["un"].write(E.NS_BUF_RECORDS)

                def drain():
                    for s in sinks.values():
                        s.drain()
                    for key in live:
                        for sk, fd in live[key]:
                            sk.close(); os.close(fd)
                        live[key] = []
                for i in range(2):         # two untimed steps back to back: the second one lands in (and sizes) the second result slot
                    w.step(998 + i, n, n_al, n_un, errlog=errlog, after_aligned=after_al, after_unaligned=after_un)
                drain()
                for e in w.engs:
                    e.io_counters(reset=True)
                t0 = time.perf_counter()
                for i in range(steps):
                    w.step(1001 + i, n, n_al, n_un, errlog=errlog, after_aligned=after_al, after_unaligned=after_un)
                t_gen = time.perf_counter() - t0
                drain()
                dt = time.perf_counter() - t0
                io = [e.io_counters() for e in w.engs]
                moved = sum(c["bytes"] for c in io)
                dma_ms = sum(c["dma_ms"] for c in io)
                res[name] = {"reads_per_s": n * steps / dt, "file_gb_per_s": moved / dt / 1e9, "seconds": dt, "bytes": moved,
                             "files": max(counter[0] * (2 if errlog else 1), len(sinks)), "host_returned_after_s": t_gen,
                             "d2h_gb_per_s_while_copying": moved / (dma_ms * 1e-3) / 1e9 if dma_ms else None,
                             "copier_waited_for_staging_s": sum(c["wait_staging_s"] for c in io), "writers_in_pwrite_s": sum(c["write_s"] for c in io)}
            finally:
                for s in sinks.values():
                    try:
                        s.close()
                    except Exception:
                        pass
                for key in live:
                    for sk, fd in live[key]:
                        try:
                            sk.close()
                        except Exception:
                            pass
                        os.close(fd)
                for fd in fds:
                    os.close(fd)
                if d is not None:
                    shutil.rmtree(d, ignore_errors=True)
    c = w.eng.io_counters()
    res["protocol"] = ("%d steps of %d reads (= %d aligned + %d unaligned) per leg after two untimed steps; record image (+ error-profile image) of every "
                       "worker call queued with ns_sink_write / ns_sink_write_range, wall time incl. the final drain; %d staging slices of %d MB, %d writer "
                       "threads per engine context, one writer per file at a time; null = /dev/null, shm = one file per output on %s (writes into ONE "
                       "inode serialise in the kernel: that bounds those legs), shm%d = %d sub-files per worker call and output, cut at read boundaries"
                       % (steps, n, n_al, n_un, c["n_slices"], c["slice_bytes"] >> 20, c["n_threads"], shm_dir, stripes, stripes))
    return res


def d2h_rate(w, nbytes=2 << 30):
    """page-locked device-to-host rate through the same pipeline with the file writes switched off (a sink without a descriptor)"""
    E = w.engine
    b = w.eng.generate(E.make_params(seed=SEED, first_read=0, n_reads=max(1000, nbytes // 9000), fastq=w.fastq, max_len=w.max_len, kmer_bias=w.kmer))
    s = w.eng.sink(-1)
    try:
        s.write(E.NS_BUF_RECORDS); s.drain()
        w.eng.io_counters(reset=True)
        t0 = time.perf_counter()
        for _ in range(4):
            s.write(E.NS_BUF_RECORDS)
        s.drain()
        dt = time.perf_counter() - t0
        c = w.eng.io_counters(reset=True)
    finally:
        s.close()
    del b
    return {"wall_gb_per_s": c["bytes"] / dt / 1e9, "dma_gb_per_s": c["d2h_gbs"], "bytes": c["bytes"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=1_000_000, help="reads per GPU per step (aligned + unaligned)")
    ap.add_argument("--serial", action="store_true", help="aligned and unaligned worker call of a step one after the other on ONE engine context")
    ap.add_argument("--aligned-only", action="store_true",
                    help="a step = one aligned worker batch only (the path with the Markov error model; profiling / A-B runs)")
    ap.add_argument("--fastq", action="store_true")
    ap.add_argument("--kmer-bias", type=int, default=0, help="-hp -k K: homopolymer expansion/contraction (configs[2] uses --fastq --kmer-bias 5)")
    ap.add_argument("--genome", choices=("ecoli", "chr1"), default="ecoli",
                    help="ecoli: 4.64 Mb circular (configs[1], the default and the headline); chr1: 248.96 Mb linear (configs[2], with --fastq --kmer-bias 5)")
    ap.add_argument("--errlog", action="store_true", help="also format the error profile on the device")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-genome-run", action="store_true", help="same as --aligned-only (kept for the profiling scripts)")
    ap.add_argument("--cpu-sample", type=int, default=5000, help="reads PER CORE of the CPU baseline sample (about 10 s with every core busy)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) or gloo (single-GPU test of the N>1 path)")
    ap.add_argument("--unaligned-delay-ms", type=float, default=0.0, help="start the unaligned worker call of a step this long after the aligned one")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end legs (generation + D2H + file writes)")
    ap.add_argument("--e2e-steps", type=int, default=3)
    ap.add_argument("--e2e-dir", default="/dev/shm")
    ap.add_argument("--no-configs2", action="store_true", help="skip the configs[2] object (chr1-size reference, FASTQ, -hp -k 5)")
    ap.add_argument("--configs2-steps", type=int, default=3)
    a = ap.parse_args()
    a.aligned_only = a.aligned_only or a.no_genome_run
    default_cfg = a.genome == "ecoli" and not a.fastq and not a.kmer_bias and not a.aligned_only and not a.serial

    import fcntl
    import numpy as np
    import torch
    import __graft_entry__ as graft
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if os.environ.get("NS_BENCH_DEVICE") is not None:          # test aid: several ranks on one GPU (gloo only)
        local_rank = int(os.environ["NS_BENCH_DEVICE"])
    # one rank builds a missing engine, the others wait on the lock and then find it up to date
    with open(os.path.join(ROOT, ".build.lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        try:
            if not os.path.exists(graft.HIP_OUT):
                graft.build()
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)
    from nanosim_amd import engine

    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if a.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # nccl backend == RCCL on ROCm
        else:
            dist.init_process_group(a.dist_backend)
    torch.cuda.set_device(local_rank)

    # ---- inputs: synthetic hg002-like model (every rank, identical by seed) + E. coli-like reference ----
    tmp = tempfile.mkdtemp(prefix="nsbench_%d_" % rank)
    w = Workload(a, a.genome, a.fastq, a.kmer_bias, local_rank, rank, world, dist, a.serial, a.aligned_only, tmp)
    n = a.reads
    n_al, n_un = w.split(n, a.aligned_only)
    infos, dt, dt_local = timed_steps(w, a, n, n_al, n_un, a.steps, a.warmup, dist, a.errlog)
    tot_bases = sum(int(x.total_bases) for st in infos for x in st)
    per_rank = None
    if dist is not None:
        rdev = "cuda" if a.dist_backend == "nccl" else "cpu"
        t = torch.tensor([dt], dtype=torch.float64, device=rdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        tb = torch.tensor([tot_bases], dtype=torch.float64, device=rdev)
        dist.all_reduce(tb, op=dist.ReduceOp.SUM)
        tot_bases = float(tb.item())
        mine = torch.tensor([dt_local / a.steps * 1e3, float(np.mean([st[0].ms_total for st in infos]))], dtype=torch.float64, device=rdev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [{"rank": r, "ms_per_step": float(x[0].item()), "aligned_device_ms": float(x[1].item())} for r, x in enumerate(allr)]

    if rank == 0:
        out = summarise(w, a, infos, dt, n, n_al, n_un, a.steps, a.warmup, world, tot_bases, a.errlog)
        if world > 1:
            out["multi_gpu"] = {"world_size": dist.get_world_size(), "backend": "RCCL (torch.distributed nccl)" if a.dist_backend == "nccl" else a.dist_backend,
                                "reference_broadcast_ms": w.broadcast_ms, "reference_bytes": w.glen, "per_rank": per_rank,
                                "collectives_in_timed_region": 0}
        if world == 1 and not a.no_e2e and w.eng_un is not None:
            try:
                out["e2e"] = {"d2h_pinned": d2h_rate(w), **e2e_legs(w, n, a.e2e_steps, a.e2e_dir)}
            except Exception as ex:                 # the headline must not depend on the state of /dev/shm
                out["e2e"] = {"error": repr(ex)}
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(w.mdl, w.ref_host, engine, a.cpu_sample, a.fastq, a.kmer_bias)
    w.close()
    if rank == 0 and world == 1 and default_cfg and not a.no_configs2:
        # BASELINE configs[2] on the same GPU, same protocol: chr1-size linear reference, FASTQ + base qualities + homopolymers
        try:
            w2 = Workload(a, "chr1", True, 5, local_rank, rank, world, None, False, False, tmp)
            infos2, dt2, _ = timed_steps(w2, a, n, n_al, n_un, a.configs2_steps, 1, None, False)
            c2 = summarise(w2, a, infos2, dt2, n, n_al, n_un, a.configs2_steps, 1, 1, sum(int(x.total_bases) for st in infos2 for x in st), False)
            out["configs2"] = {k: c2[k] for k in ("value", "unit", "bases_per_s", "steps", "warmup", "ms_per_step", "config", "device_ms_per_step",
                                                  "aligned_batch", "roofline", "unaligned_batch")}
            w2.close()
        except Exception as ex:
            out["configs2"] = {"error": repr(ex)}
    if rank == 0:
        print(json.dumps(out))
    shutil.rmtree(tmp, ignore_errors=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
