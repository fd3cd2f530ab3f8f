#!/usr/bin/env python
"""bench.py — simulated reads/s (+ bases/s) of the genome-mode hot path on N MI355X.

A "step" = one ns_generate() pass (one worker call of the reference, src/simulator.py:1266-1454) over one
batch of reads of BASELINE.json configs[1]: E. coli-like 4.64 Mb circular genome, hg002-like error model
(mean aligned length ~8.4 kb, ~265 error events/read), FASTA records, reference + model resident in HBM,
outputs left in HBM.  N>1: one process per GPU, read-index ranges sharded, ONE RCCL broadcast of the
reference before the timed region, no collective inside it (weak scaling).

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


def cpu_baseline(model, ref, engine_mod, n_sample):
    """The CPU restatement (oracle, kind="port") on ONE host core, bounded sample of the same workload."""
    from tests import oracle_lib
    p = engine_mod.make_params(seed=SEED, first_read=0, n_reads=n_sample, max_len=ref.max_chrom)
    t0 = time.perf_counter()
    out = oracle_lib.generate(model, ref, p, bytes_per_read=60000)
    dt = time.perf_counter() - t0
    return dict(value=n_sample / dt, unit="reads/s", cores=1, kind="port",
                sample="%d aligned reads of the same workload, oracle/ns_oracle.c on 1 core, %.1f s" % (n_sample, dt),
                bases_per_s=out["total_bases"] / dt)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=1_000_000, help="reads per GPU per step")
    ap.add_argument("--engines", type=int, default=int(os.environ.get("NS_BENCH_ENGINES", "1")),
                    help="engine contexts per GPU, each driven by its own host thread: the Markov-chain stage of one batch "
                         "overlaps the record stage of another")
    ap.add_argument("--fastq", action="store_true")
    ap.add_argument("--kmer-bias", type=int, default=0, help="-hp -k K: homopolymer expansion/contraction (configs[2] uses --fastq --kmer-bias 5)")
    ap.add_argument("--genome", choices=("ecoli", "chr1"), default="ecoli",
                    help="ecoli: 4.64 Mb circular (configs[1], the default and the headline); chr1: 248.96 Mb linear (configs[2], with --fastq --kmer-bias 5)")
    ap.add_argument("--errlog", action="store_true", help="also format the error profile on the device")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=40000)
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) or gloo (single-GPU test of the N>1 path)")
    a = ap.parse_args()

    import numpy as np
    import torch
    import __graft_entry__ as graft
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if os.environ.get("NS_BENCH_DEVICE") is not None:          # test aid: several ranks on one GPU (gloo only)
        local_rank = int(os.environ["NS_BENCH_DEVICE"])
    if not os.path.exists(graft.HIP_OUT):
        if rank == 0:
            graft.build()
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
    engs = [engine.Engine(local_rank) for _ in range(max(1, a.engines))]
    eng = engs[0]
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

    n = a.reads
    def step(i, e=None):
        p = engine.make_params(seed=SEED, first_read=(i * world + rank) * n, n_reads=n, fastq=a.fastq,
                               max_len=min(glen, 1 << 30), emit_errlog=a.errlog, kmer_bias=a.kmer_bias)
        return (e or eng).generate(p)

    def run_steps(first, count):
        """`count` steps; with several engines, engine k takes steps k, k + E, ... in its own host thread (the C call releases the GIL)"""
        if len(engs) == 1:
            return [step(first + i).info for i in range(count)]
        import threading
        infos = [None] * count
        def work(k):
            for i in range(k, count, len(engs)):
                infos[i] = step(first + i, engs[k]).info
        th = [threading.Thread(target=work, args=(k,)) for k in range(len(engs))]
        for t in th:
            t.start()
        for t in th:
            t.join()
        return infos

    run_steps(0, a.warmup)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    infos = run_steps(a.warmup, a.steps)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    tot_bases = sum(int(x.total_bases) for x in infos)
    if dist is not None:
        rdev = "cuda" if a.dist_backend == "nccl" else "cpu"
        t = torch.tensor([dt], dtype=torch.float64, device=rdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        tb = torch.tensor([tot_bases], dtype=torch.float64, device=rdev)
        dist.all_reduce(tb, op=dist.ReduceOp.SUM)
        tot_bases = float(tb.item())

    if rank == 0:
        # roofline of the dominant kernel: algorithmic bytes per launch / its mean HIP-event duration
        kms = {}
        for k, nm in enumerate(engine.KERNEL_NAMES):
            kms[nm] = float(np.mean([x.ms_kernel[k] for x in infos]))
        dom = max(kms, key=kms.get)
        per_launch = np.mean([int(x.total_ref_bases) + int(x.total_bases) * (2 if a.fastq else 1) +
                              16 * int(x.events_used) + 32 * int(x.n_reads) for x in infos])
        achieved = per_launch / (kms[dom] * 1e-3) / 1e9
        device_ms = float(np.mean([x.ms_total for x in infos]))
        # HBM traffic of the dominant kernel: rocprofv3 PMC passes (FETCH_SIZE x2 per the gfx950 note + WRITE_SIZE) cannot be
        # collected from inside this process; the per-read figure measured by scripts/profile_round.sh is committed in
        # profiles/r01/pmc_summary.json and scaled to this launch size.  null when that file has no entry for the kernel.
        traffic = None
        stage = {"k_materialise": ("k_words", "k_materialise")}.get(dom, (dom,))     # kernels behind the timed stage
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "r01", "pmc_summary.json")))
            tot = 0.0
            for kname, kv in pm["kernels"].items():
                if kname.startswith(stage) and "hbm_bytes_per_read" in kv:
                    tot += kv["hbm_bytes_per_read"]
            if tot > 0 and not a.fastq and not a.kmer_bias and a.genome == "ecoli":      # (the committed counters are for the default workload)
                traffic = tot * n
        except (OSError, ValueError, KeyError):
            traffic = None
        out = {
            "metric": "simulated reads/sec (genome mode, mean 8 kb)", "value": world * n * a.steps / dt, "unit": "reads/s",
            "bases_per_s": tot_bases / dt,
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": ("configs[1]: ecoli_like 4,641,652 bp circular" if a.genome == "ecoli" else "configs[2]: chr1_like 248,956,422 bp linear") +
                                   ", hg002_like error model, genome mode, %s%s, %d reads/GPU/step" % ("FASTQ" if a.fastq else "FASTA", ", -hp -k %d" % a.kmer_bias if a.kmer_bias else "", n),
                       "reads_per_step_per_gpu": n, "errlog": bool(a.errlog), "seed": SEED,
                       "parallelism": "read-index sharding x%d, 1 RCCL broadcast of the reference" % world, "engines_per_gpu": len(engs)},
            "device_ms_per_step": device_ms, "kernel_ms": kms,
            "roofline": {"bound": "hbm", "kernel": dom + (" (stage: k_words + k_materialise + k_materialise_slow)" if dom == "k_materialise" else ""), "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": "profiles/r01/pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, per read x reads per launch)",
                         "algorithmic_bytes_per_launch": float(per_launch),
                         "all_kernels_achieved": per_launch / (sum(kms.values()) * 1e-3) / 1e9},
        }
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(mdl, ref_host, engine, a.cpu_sample)
        print(json.dumps(out))
    for e in engs:
        e.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
