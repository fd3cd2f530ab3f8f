#!/usr/bin/env python
"""bench.py — simulated reads/s (+ bases/s) of the genome-mode hot path on N MI355X.

A "step" = one genome-mode pass over one batch of read indices of BASELINE.json configs[1]: the aligned worker
call (src/simulator.py:1266-1454) on 950 000 reads and the unaligned one (S:1482-1549) on 50 000 (the model's
alignment rate 19:1), E. coli-like 4.64 Mb circular genome, hg002-like error model (mean aligned length ~8.4 kb,
~265 error events/read), FASTA records, reference + model resident in HBM, outputs left in HBM; the two calls run side by side on two engine contexts of the GPU.  N>1: one
process per GPU, read-index ranges sharded, ONE RCCL broadcast of the reference before the timed region, no
collective inside it (weak scaling).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--reads R]
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SEED = 20260926
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


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


def measured_traffic(a, kernels):
    """HBM bytes per read of the kernels behind the roofline stage, from the rocprofv3 PMC passes of THIS configuration
    (scripts/profile_round.sh -> profiles/r02/pmc_<config>.json: FETCH_SIZE x 2 per the gfx950 note + WRITE_SIZE, separate passes).
    PMC counters cannot be collected from inside this process; None when the configuration has not been profiled."""
    key = "%s_%s%s" % (a.genome, "fastq" if a.fastq else "fasta", "_k%d" % a.kmer_bias if a.kmer_bias else "")
    path = os.path.join(ROOT, "profiles", "r02", "pmc_%s.json" % key)
    try:
        pm = json.load(open(path))
    except (OSError, ValueError):
        return None, None
    tot = 0.0
    for kname, kv in pm.get("kernels", {}).items():
        if kname.startswith(kernels) and "hbm_bytes_per_read" in kv:
            tot += kv["hbm_bytes_per_read"]
    return (tot if tot > 0 else None), os.path.relpath(path, ROOT)


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
    a = ap.parse_args()
    a.aligned_only = a.aligned_only or a.no_genome_run

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
    from nanosim_amd import engine, model, synth

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
    prefix = os.path.join(tmp, "hg002_like")
    synth.write_model(prefix, synth.SynthModelSpec(n_train=1_000_000, seed=SEED), write_pkl=False)
    mdl = model.load_model(prefix, fastq=a.fastq, homopolymer=a.kmer_bias > 0)
    names = ["ecoli-like"] if a.genome == "ecoli" else ["chr1-like"]
    glen = synth.ECOLI_LEN if a.genome == "ecoli" else synth.CHR1_LEN
    ref_meta = model.Reference(names, np.zeros(0, np.uint8), np.array([0, glen], dtype=np.uint64),
                               np.array([1 if a.genome == "ecoli" else 0], dtype=np.uint8))
    eng = engine.Engine(local_rank)
    # the unaligned worker call of a step runs next to the aligned one on its own engine context (own HIP streams and buffers on the
    # same GPU, own host thread) — the way the reference runs its workers side by side (-t, S:1588-1605)
    eng_un = None if (a.aligned_only or a.serial) else engine.Engine(local_rank)
    if eng_un is not None:
        eng_un.set_background(True)     # its kernels share the GPU with the aligned call's: few issue slots matter more than a short latency
    engs = [e for e in (eng, eng_un) if e is not None]
    if world > 1:
        # the reference lives on rank 0; ONE broadcast over xGMI puts it in every GPU's HBM
        bdev = "cuda" if a.dist_backend == "nccl" else "cpu"
        buf = torch.empty(glen, dtype=torch.uint8, device=bdev)
        if rank == 0:
            seq = synth.synth_sequence(glen, SEED, n_frac=0.0005, iupac_frac=0.0002, lower_frac=0.02, hp_boost=0.005)
            buf.copy_(torch.from_numpy(seq))
        dist.broadcast(buf, src=0)
        buf = buf.cuda()
        torch.cuda.synchronize()
        for e in engs:
            e.set_reference_device(buf.data_ptr(), ref_meta)
        ref_host = None
    else:
        seq = synth.synth_sequence(glen, SEED, n_frac=0.0005, iupac_frac=0.0002, lower_frac=0.02, hp_boost=0.005)
        ref_host = model.Reference(names, seq, ref_meta.chrom_off, ref_meta.circular)
        for e in engs:
            e.set_reference(ref_host)
    for e in engs:
        e.load_model(mdl)

    # ---- a step = one genome-mode pass of this GPU over n read indices: the aligned worker call (simulation_aligned_genome,
    # S:1266-1454) on round(n r / (r + 1)) reads, then the unaligned one (simulation_unaligned, S:1482-1549) on the rest
    # (the model's alignment rate r = 19), as simulation() runs them (S:1571-1672)
    n = a.reads
    n_al, n_un = (n, 0) if a.aligned_only else mdl.split_counts(n)
    max_len = min(glen, 1 << 30)

    import threading

    def step(i):
        base = (i * world + rank) * n
        out = [None, None]

        def aligned():
            out[0] = eng.generate(engine.make_params(seed=SEED, first_read=base, n_reads=n_al, fastq=a.fastq, max_len=max_len,
                                                     emit_errlog=a.errlog, kmer_bias=a.kmer_bias)).info

        def unaligned(e):
            out[1] = e.generate(engine.make_params(seed=SEED, first_read=base + n_al, n_reads=n_un, kind=engine.NS_KIND_UNALIGNED,
                                                   fastq=a.fastq, max_len=max_len)).info
        if not n_un:
            aligned()
            return out[:1]
        if eng_un is None:                                # --serial: one engine, one call after the other
            aligned(); unaligned(eng)
            return out
        t = threading.Thread(target=unaligned, args=(eng_un,))      # (the C call releases the GIL)
        t.start(); aligned(); t.join()
        return out

    for i in range(a.warmup):
        step(i)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    infos = [step(a.warmup + i) for i in range(a.steps)]
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    tot_bases = sum(int(x.total_bases) for st in infos for x in st)
    if dist is not None:
        rdev = "cuda" if a.dist_backend == "nccl" else "cpu"
        t = torch.tensor([dt], dtype=torch.float64, device=rdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        tb = torch.tensor([tot_bases], dtype=torch.float64, device=rdev)
        dist.all_reduce(tb, op=dist.ReduceOp.SUM)
        tot_bases = float(tb.item())

    if rank == 0:
        al = [st[0] for st in infos]
        # roofline of the dominant kernel (of the aligned worker call): algorithmic bytes per launch / its mean HIP-event duration
        kms = {}
        for k, nm in enumerate(engine.KERNEL_NAMES):
            kms[nm] = float(np.mean([x.ms_kernel[k] for x in al]))
        dom = max(kms, key=kms.get)
        per_launch = np.mean([int(x.total_ref_bases) + int(x.total_bases) * (2 if a.fastq else 1) +
                              16 * int(x.events_used) + 32 * int(x.n_reads) for x in al])
        achieved = per_launch / (kms[dom] * 1e-3) / 1e9
        device_ms = float(np.mean([sum(x.ms_total for x in st) for st in infos]))
        stage = {"k_materialise": ("k_words", "k_materialise"), "k_hp": ("k_hp", "k_words", "k_materialise<true, 1>", "k_materialise<false, 1>")}.get(dom, (dom,))
        per_read, traffic_src = measured_traffic(a, stage)
        out = {
            "metric": "simulated reads/sec (genome mode, mean 8 kb)", "value": world * n * a.steps / dt, "unit": "reads/s",
            "bases_per_s": tot_bases / dt,
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": ("configs[1]: ecoli_like 4,641,652 bp circular" if a.genome == "ecoli" else "configs[2]: chr1_like 248,956,422 bp linear") +
                                   ", hg002_like error model, genome mode, %s%s, %d reads/GPU/step = %d aligned + %d unaligned (alignment rate 19:1)"
                                   % ("FASTQ" if a.fastq else "FASTA", ", -hp -k %d" % a.kmer_bias if a.kmer_bias else "", n, n_al, n_un),
                       "reads_per_step_per_gpu": n, "aligned_per_step": n_al, "unaligned_per_step": n_un, "errlog": bool(a.errlog),
                       "errlog_note": "the error-profile text (the reference always writes it, S:2006-2008: ~26 KB per read, 3x the reads) is "
                                      "formatted by k_errlog only when asked for (--errlog; the CLI always asks): it is a file-format stage "
                                      "behind the path the metric names (SURVEY section 8 f-1)",
                       "seed": SEED, "parallelism": "read-index sharding x%d, 1 RCCL broadcast of the reference" % world, "engines_per_gpu": len(engs),
                       "engines_note": "aligned and unaligned worker call of a step run side by side on two engine contexts of the GPU, the unaligned one as a background context (ns_set_background); --serial: one after the other on one"},
            "device_ms_per_step": device_ms,
            "aligned_batch": {"reads": n_al, "device_ms": float(np.mean([x.ms_total for x in al])),
                              "reads_per_s_device": n_al / (float(np.mean([x.ms_total for x in al])) * 1e-3), "kernel_ms": kms},
            "kernel_ms": kms,
            "roofline": {"bound": "hbm", "kernel": dom + (" (stage: k_words + k_materialise + k_materialise_slow)" if dom == "k_materialise" else ""), "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": per_read * n_al if per_read else None,
                         "traffic_source": (traffic_src + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this configuration, per read x reads per launch)") if per_read else None,
                         "algorithmic_bytes_per_launch": float(per_launch),
                         "all_kernels_achieved": per_launch / (sum(kms.values()) * 1e-3) / 1e9},
        }
        if n_un:
            un = [st[1] for st in infos]
            out["unaligned_batch"] = {"reads": n_un, "device_ms": float(np.mean([x.ms_total for x in un])),
                                      "kernel_ms": {nm: float(np.mean([x.ms_kernel[k] for x in un])) for k, nm in enumerate(engine.KERNEL_NAMES)}}
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(mdl, ref_host, engine, a.cpu_sample, a.fastq, a.kmer_bias)
        print(json.dumps(out))
    for e in engs:
        e.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
