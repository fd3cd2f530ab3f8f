#!/usr/bin/env python
"""bench.py — simulated reads/s (+ bases/s) of the genome-mode hot path on N MI355X.

A "step" = one genome-mode pass over one batch of read indices of BASELINE.json configs[1]: the aligned worker
call (src/simulator.py:1266-1454) on 950 000 reads and the unaligned one (S:1482-1549) on 50 000 (the model's
alignment rate 19:1), E. coli-like 4.64 Mb circular genome, hg002-like error model (mean aligned length ~8.4 kb,
~265 error events/read), FASTA records, reference + model resident in HBM, outputs left in HBM; the two calls run side by side on two engine contexts of the GPU.  N>1: one
process per GPU, read-index ranges sharded, ONE RCCL broadcast of the reference before the timed region, no
collective inside it (weak scaling).

Other workloads (not the headline; one line each): --genome chr1 --fastq --kmer-bias 5 = configs[2]; --genome grch38 --chimeric =
configs[3] (3.1 Gb reference, chimeric reads: the 8-GPU genome split); --metagenome = configs[4] (zymo10-like community of 10 species,
one metagenome worker call per step).  --gpus N without a launcher starts the N ranks itself (torch.distributed.run on 127.0.0.1).

The same JSON line also carries (N = 1):
  "serial"    the same steps with both worker calls one after the other on ONE engine context (NS_SERIAL=1 of the CLI);
  "errlog_on" the same steps with the error profile the reference always writes (S:2006-2008) formatted on the device as well — what
              the CLI's worker calls cost on the device — and k_errlog's own store rate as a fraction of HBM peak;
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
PROFILE_ROUNDS = ("r06", "r05", "r04", "r03", "r02")


REFERENCE_PYTHON = {        # the reference itself (bcgsc/NanoSim v3.2.2, simulator.py -t 8), measured in the build container
    "reads_per_s_per_core": {"fasta": 190, "fastq_hp_k5": 44},
    "hardware": "8 vCPU Intel Xeon @ 2.10 GHz (build container, not the GPU box), Python 3.10 / numpy 2.2",
    "note": "the reference's Python cannot travel to the GPU box; its C restatement (oracle/) is timed there instead",
}


def _cpu_worker(args):
    """one host core: its share of the sample through the C restatement of the reference (oracle/ns_oracle.c)"""
    idx, n_al, n_un, fastq, kmer, chimeric = args
    from tests import oracle_lib
    mdl, ref, eng = _CPU_CTX
    bases = 0
    first = idx * (n_al + n_un)
    for kind, cnt in ((eng.NS_KIND_ALIGNED, n_al), (eng.NS_KIND_UNALIGNED, n_un)):
        done = 0
        while done < cnt:                            # 1 000 reads per call: the buffers of a call stay below ~50 MB per process
            m = min(1000, cnt - done)
            p = eng.make_params(seed=SEED, first_read=first, n_reads=m, kind=kind, max_len=ref.max_chrom, fastq=fastq,
                                kmer_bias=kmer if kind == eng.NS_KIND_ALIGNED else 0, chimeric=chimeric and kind == eng.NS_KIND_ALIGNED)
            per = max(120000 if fastq else 60000, 8_000_000 // m)     # a single FASTQ record of a long read needs more than the average
            bases += int(oracle_lib.generate(mdl, ref, p, bytes_per_read=per)["total_bases"])
            done += m; first += m
    return bases


_CPU_CTX = None


def cpu_baseline(model, ref, engine_mod, per_core, fastq, kmer, chimeric=False):
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
        pool.map(_cpu_worker, [(i, 20, 0, fastq, kmer, chimeric) for i in range(cores)])         # start the workers, touch the tables
        t0 = time.perf_counter()
        bases = pool.map(_cpu_worker, [(i, n_al, n_un, fastq, kmer, chimeric) for i in range(cores)])
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


WORKLOADS = {
    "ecoli": "configs[1]: ecoli_like 4,641,652 bp circular",
    "chr1": "configs[2]: chr1_like 248,956,422 bp linear",
    "grch38": "configs[3]: grch38_like 24 chromosomes, 3,088,269,832 bp linear",
    "zymo10": "configs[4]: zymo10_like community (10 species, 44 chromosomes, 8 circular bacteria + 2 yeasts), even abundances, metagenome mode",
}


def reference_layout(genome):
    """(names, chrom_off, circular) of a synthetic reference — every rank knows the layout, only rank 0 makes the bases"""
    import numpy as np
    from nanosim_amd import synth
    if genome == "ecoli":
        return ["ecoli-like"], np.array([0, synth.ECOLI_LEN], dtype=np.uint64), np.array([1], dtype=np.uint8)
    if genome == "chr1":
        return ["chr1-like"], np.array([0, synth.CHR1_LEN], dtype=np.uint64), np.array([0], dtype=np.uint8)
    if genome == "grch38":
        names = ["chr%d" % (i + 1) for i in range(22)] + ["chrX", "chrY"]
        off = np.concatenate([[0], np.cumsum(np.array(synth.GRCH38_LENS, dtype=np.uint64))]).astype(np.uint64)
        return names, off, np.zeros(24, dtype=np.uint8)
    names, lens, circ = [], [], []
    for sp, chroms, _, _ in synth.ZYMO10:
        for k, n, c in chroms:
            names.append(sp + "-" + k); lens.append(int(n)); circ.append(c)
    off = np.concatenate([[0], np.cumsum(np.array(lens, dtype=np.uint64))]).astype(np.uint64)
    return names, off, np.array(circ, dtype=np.uint8)


_REF_CACHE = {}


def reference_bases(genome):
    """the synthetic reference of a workload (cached: the configs[2] and chr1 FASTA objects of the default line share one)"""
    if genome not in _REF_CACHE:
        _REF_CACHE.clear()                         # one at a time: grch38 is 3.1 GB
        _REF_CACHE[genome] = _reference_bases(genome)
    return _REF_CACHE[genome]


def _reference_bases(genome):
    from nanosim_amd import synth
    kw = dict(n_frac=0.0005, iupac_frac=0.0002, lower_frac=0.02, hp_boost=0.005)
    if genome == "ecoli":
        return synth.synth_sequence(synth.ECOLI_LEN, SEED, **kw)
    if genome == "chr1":
        return synth.synth_sequence(synth.CHR1_LEN, SEED, **kw)
    if genome == "grch38":
        return synth.grch38_like(SEED)[1]
    return synth.zymo10_like(SEED)[1]


class Workload:
    """one configuration resident on this rank's GPU: engines (aligned + background unaligned context), model, reference"""

    def __init__(self, a, genome, fastq, kmer, local_rank, rank, world, dist, serial, aligned_only, tmp, chimeric=False):
        import numpy as np
        import torch
        from nanosim_amd import engine, model, synth
        self.engine, self.genome, self.fastq, self.kmer, self.rank, self.world = engine, genome, fastq, kmer, rank, world
        self.meta = genome == "zymo10"
        self.chimeric = bool(chimeric)
        self.trained_shape = bool(getattr(a, "trained_shape", False))
        prefix = os.path.join(tmp, "hg002_like" + ("_trained_shape" if self.trained_shape else ""))
        if not os.path.exists(prefix + "_kde.npz"):
            spec = synth.SynthModelSpec(n_train=1_000_000, seed=SEED)
            if self.trained_shape:
                # the table SHAPE of a model read_analysis.py trains (README.md:41: the published ones are not in this image): 15 previous-match
                # bins and 1 500-row ECDFs — ~360 KB of chain tables, far beyond the LDS image, so k_chain reads them from global memory
                # (k_chain<false, false>, ns_chain.h: chain_error_list_g).  Same error rates as hg002_like.
                bins = ((0, 1), (1, 2), (2, 3), (3, 5), (5, 7), (7, 10), (10, 14), (14, 19), (19, 25), (25, 33), (33, 45), (45, 60),
                        (60, 90), (90, 150), (150, 1500))
                means = (24.0, 25.0, 26.0, 27.0, 28.0, 29.0, 30.0, 31.0, 31.0, 32.0, 33.0, 34.0, 35.0, 36.0, 36.0)
                spec = synth.SynthModelSpec(n_train=1_000_000, seed=SEED, ecdf_rows=1500, mm_bins=bins, mm_means=means, mm_zero=(0.0,) + (0.03,) * 14)
            synth.write_model(prefix, spec, write_pkl=False)
        self.mdl = model.load_model(prefix, fastq=fastq, homopolymer=kmer > 0, chimeric=self.chimeric)
        names, chrom_off, circular = reference_layout(genome)
        self.glen = glen = int(chrom_off[-1])
        ref_meta = model.Reference(names, np.zeros(0, np.uint8), chrom_off, circular)
        self.eng = engine.Engine(local_rank)
        # the unaligned worker call of a step runs next to the aligned one on its own engine context (own HIP streams and buffers on the
        # same GPU, own host thread): the schedule of the CLI (nanosim_amd/simulator.py: _run_phases)
        # default: ns_generate_step — the library runs the unaligned call on the engine's step companion (which shares reference and model)
        # from its own worker thread; --python-threads: the round-2..4 form, a second Engine with its own copy of everything and a
        # Python thread per step (kept for A/B runs)
        self.python_threads = bool(getattr(a, "python_threads", False))
        if aligned_only or serial:
            self.eng_un = None
        elif self.python_threads:
            self.eng_un = engine.Engine(local_rank)
            self.eng_un.set_background(True)     # (a scheduling hint: its kernels share the GPU with the aligned call's)
        else:
            self.eng_un = self.eng.step_engine()
        self.own_engs = [e for e in (self.eng, self.eng_un if self.python_threads else None) if e is not None]      # what this object sets up
        self.engs = [e for e in (self.eng, self.eng_un) if e is not None]
        self.broadcast_ms = None
        dev_ptr, self._keep = None, None
        if world > 1:
            # the reference lives on rank 0; ONE broadcast over xGMI puts it in every GPU's HBM
            bdev = "cuda" if a.dist_backend == "nccl" else "cpu"
            buf = torch.empty(glen, dtype=torch.uint8, device=bdev)
            if rank == 0:
                buf.copy_(torch.from_numpy(reference_bases(genome)))
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            dist.broadcast(buf, src=0)
            torch.cuda.synchronize()
            self.broadcast_ms = (time.perf_counter() - t0) * 1e3
            buf = buf.cuda()
            torch.cuda.synchronize()
            dev_ptr, self._keep = buf.data_ptr(), buf
            self.ref_host = None
            ref = ref_meta
        else:
            self.ref_host = ref = model.Reference(names, reference_bases(genome), chrom_off, circular)
        if self.meta:
            from nanosim_amd import metagenome as MG
            sp_off = np.concatenate([[0], np.cumsum([len(c) for _, c, _, _ in synth.ZYMO10])]).astype(np.uint32)
            self.mref = MG.MetaReference(ref, [sp for sp, _, _, _ in synth.ZYMO10], sp_off, [[k for k, _, _ in c] for _, c, _, _ in synth.ZYMO10])
            abun = {sp: float(e) for sp, _, e, _ in synth.ZYMO10}                  # the "even" column of the abundance table
            infl = {sp: MG.inflate_abun(abun, sp, self.mdl.abun_inflation) for sp in abun} if self.chimeric else None
            self.eng.set_metagenome(self.mref, abun, infl, dev_ptr=dev_ptr)
            if self.eng_un is not None and self.python_threads:
                self.eng_un.set_metagenome(self.mref, dev_ptr=dev_ptr)
        else:
            for e in self.own_engs:
                if dev_ptr is not None:
                    e.set_reference_device(dev_ptr, ref_meta)
                else:
                    e.set_reference(ref)
        self._keep = None                          # (the engines hold their own normalised copies)
        for e in self.own_engs:
            e.load_model(self.mdl)
        self.max_len = min(int(np.diff(chrom_off.astype(np.int64)).max()), 1 << 30)

    def describe(self):
        mode = "metagenome mode" if self.meta else "genome mode"
        return (WORKLOADS[self.genome] + ", hg002_like error model" + (" in the table shape of a trained model (15 bins x 1 500 rows: chain tables in global memory)" if self.trained_shape else "") +
                ("" if self.meta else ", " + mode) + ", " + ("FASTQ" if self.fastq else "FASTA") +
                (", -hp -k %d" % self.kmer if self.kmer else "") + (", --chimeric" if self.chimeric else ""))

    def split(self, n, aligned_only):
        return (n, 0) if aligned_only else self.mdl.split_counts(n)

    def step(self, i, n, n_al, n_un, errlog=False, records=True, after_aligned=None, after_unaligned=None, serial=False):
        """a step = one pass of this GPU over n read indices: the aligned worker call (simulation_aligned_genome, S:1266-1454;
        --metagenome: simulation_aligned_metagenome, S:814-1040) on round(n r / (r + 1)) reads and the unaligned one
        (simulation_unaligned, S:1482-1549) on the rest (the model's alignment rate r = 19), as simulation() runs them (S:1571-1672).
        after_*: called with the batch right after its worker call (the end-to-end legs queue the result buffers for their files there).
        serial: both calls on the first engine context, one after the other."""
        engine = self.engine
        base = (i * self.world + self.rank) * n
        out = [None, None]

        p_al = engine.make_params(seed=SEED, first_read=base, n_reads=n_al, fastq=self.fastq, max_len=self.max_len, emit_errlog=errlog,
                                  kmer_bias=self.kmer, emit_records=records, chimeric=self.chimeric, meta=self.meta)
        p_un = engine.make_params(seed=SEED, first_read=base + n_al, n_reads=n_un, kind=engine.NS_KIND_UNALIGNED, fastq=self.fastq,
                                  max_len=self.max_len, emit_records=records, meta=self.meta)

        def aligned():
            b = self.eng.generate(p_al)
            out[0] = b.info
            if after_aligned:
                after_aligned(b)

        def unaligned(e):
            b = e.generate(p_un)
            out[1] = b.info
            if after_unaligned:
                after_unaligned(b)
        if not n_un:
            aligned()
            return out[:1]
        if self.eng_un is None or serial:                      # one engine, one call after the other
            aligned(); unaligned(self.eng)
            return out
        if not self.python_threads:                            # ONE library call per step (ns_generate_step)
            b_al, b_un = self.eng.generate_step(p_al, p_un)
            out[0], out[1] = b_al.info, b_un.info
            if after_aligned:
                after_aligned(b_al)
            if after_unaligned:
                after_unaligned(b_un)
            return out
        t = threading.Thread(target=unaligned, args=(self.eng_un,))      # --python-threads (the C call releases the GIL)
        t.start(); aligned(); t.join()
        return out

    def close(self):
        for e in self.own_engs:
            e.close()


def timed_steps(w, a, n, n_al, n_un, steps, warmup, dist, errlog, serial=False, first_step=0):
    import torch
    for i in range(warmup):
        w.step(first_step + i, n, n_al, n_un, errlog=errlog, serial=serial)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    infos = [w.step(first_step + warmup + i, n, n_al, n_un, errlog=errlog, serial=serial) for i in range(steps)]
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
    # ... of the kernel ITSELF when the dominant stage is the record stage: ms_kernel[6] brackets k_materialise alone (what rocprofv3's kernel
    # trace reports for it); the stage time — kms["k_materialise"]: + the memset of the slow-tile queue, the generic kernel for queued tiles,
    # the join with k_names on the second stream — stays in kernel_ms and prices frac_stage
    t_stage = kms[dom] * 1e-3
    k_only = float(np.mean([x.ms_kernel[6] for x in al])) if dom == "k_materialise" else 0.0
    t_kernel = k_only * 1e-3 if k_only > 0 else t_stage
    achieved = per_launch / t_kernel / 1e9
    device_ms = float(np.mean([sum(x.ms_total for x in st) for st in infos]))
    stage = {"k_materialise": ("k_materialise",), "k_hp": ("k_hp", "k_materialise<true, 1>", "k_materialise<false, 1>")}.get(dom, (dom,))
    per_read, traffic_src = measured_traffic(w.genome, w.fastq, w.kmer, stage)
    # the same duration priced three ways, so that the line cannot flatter itself: SURVEY 8(d) bytes (incl. the 8 B per event the CHAIN
    # kernel writes), the bytes this stage itself moves (without them), and the bytes the PMC counters saw
    ev_write = float(np.mean([8 * int(x.events_used) for x in al]))
    t_dom = t_kernel
    out = {
        "metric": "simulated reads/sec (genome mode, mean 8 kb)", "value": world * n * steps / dt, "unit": "reads/s",
        "bases_per_s": tot_bases / dt,
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": w.describe() + ", %d reads/GPU/step = %d aligned + %d unaligned (alignment rate 19:1)" % (n, n_al, n_un),
                   "reads_per_step_per_gpu": n, "aligned_per_step": n_al, "unaligned_per_step": n_un, "errlog": bool(errlog),
                   "errlog_note": "the error-profile text (the reference always writes it, S:2006-2008: ~26 KB per read, 3x the reads) is "
                                  "formatted by k_errlog only when asked for (--errlog; the CLI always asks): it is a file-format stage "
                                  "behind the path the metric names (SURVEY section 8 f-1); the e2e legs below time it",
                   "seed": SEED, "parallelism": "read-index sharding x%d, 1 RCCL broadcast of the reference" % world, "engines_per_gpu": len(w.engs),
                   "engines_note": "one ns_generate_step call per step (include/nanosim_amd.h, ABI 6): the aligned worker call on the engine context, the unaligned one on its step companion (same reference and model, own streams and batch buffers, the library's worker thread) side by side on the GPU - the schedule of the CLI (nanosim_amd/simulator.py: StepPair); --python-threads: two independent Engine objects and a Python thread per step (rounds 2-4); the `serial` object / --serial: one after the other on one context (the CLI with NS_SERIAL=1)",
                   "step_call": "python threads" if w.python_threads else "ns_generate_step"},
        "device_ms_per_step": device_ms,
        "aligned_batch": {"reads": n_al, "device_ms": float(np.mean([x.ms_total for x in al])),
                          "reads_per_s_device": n_al / (float(np.mean([x.ms_total for x in al])) * 1e-3), "kernel_ms": kms},
        "kernel_ms": kms,
        "roofline": {"bound": "hbm", "kernel": dom + (" (HIP events around the kernel itself: ns_batch_info.ms_kernel[6]; stage_ms / frac_stage: + the generic kernel for queued tiles and the join with k_names on the second stream)" if dom == "k_materialise" else ""), "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS,
                     "kernel_ms": t_kernel * 1e3, "stage_ms": t_stage * 1e3, "frac_stage": per_launch / t_stage / 1e9 / HBM_PEAK_GBS,
                     "frac_kernel_only_bytes": (per_launch - ev_write) / t_dom / 1e9 / HBM_PEAK_GBS,
                     "frac_counter_bytes": (per_read * n_al / t_dom / 1e9 / HBM_PEAK_GBS) if per_read else None,
                     "frac_note": "frac: SURVEY 8(d) algorithmic bytes (L_ref + L_out [+ L_out qualities] + 16 E + 32) / kernel time / 8 TB/s; "
                                  "frac_kernel_only_bytes: without the 8 B per event the chain kernel writes; frac_counter_bytes: PMC bytes "
                                  "(2 x FETCH_SIZE + WRITE_SIZE of the profiled launch) / kernel time / 8 TB/s",
                     "traffic": per_read * n_al if per_read else None,
                     "traffic_source": (traffic_src + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this configuration, per read x reads per launch)") if per_read else None,
                     "algorithmic_bytes_per_launch": float(per_launch),
                     "all_kernels_achieved": per_launch / (sum(kms.values()) * 1e-3) / 1e9,
                     "whole_aligned_batch_frac": per_launch / (float(np.mean([x.ms_total for x in al])) * 1e-3) / 1e9 / HBM_PEAK_GBS},
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


def _free_port():
    import socket
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    return port


def self_launch(a):
    """--gpus N without a launcher: start the N ranks here — one process per GPU under torch.distributed.run, rendezvous on 127.0.0.1 —
    with the same arguments; rank 0's JSON line is this process's output."""
    import subprocess
    if a.dist_backend == "nccl" and os.environ.get("NS_BENCH_DEVICE") is None:
        import torch
        have = torch.cuda.device_count()
        if have < a.gpus:
            sys.stderr.write("bench.py --gpus %d: %d GPU(s) visible (NS_BENCH_DEVICE=0 with --dist-backend gloo runs the %d ranks on one GPU)\n"
                             % (a.gpus, have, a.gpus))
            sys.exit(2)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC (RCCL between processes)
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


def extra_legs(w, a, n, n_al, n_un, steps):
    """N = 1: what the headline schedule leaves out.  serial: both worker calls of a step on ONE engine context, one after the other
    (S:1621-1622 joins the aligned workers before S:1642-1663 starts the unaligned ones; the CLI with NS_SERIAL=1).  errlog_on: the
    headline schedule with the error profile formatted on the device too (emit_errlog = 1: k_errlen + k_errlog), as every worker call of
    the CLI runs — the device-side rate of the CLI — with k_errlog's own store rate against the HBM peak."""
    import numpy as np
    out = {}
    infos, dt, _ = timed_steps(w, a, n, n_al, n_un, steps, 1, None, False, serial=True, first_step=2000)
    out["serial"] = {"value": n * steps / dt, "unit": "reads/s", "ms_per_step": dt / steps * 1e3, "steps": steps,
                     "aligned_device_ms": float(np.mean([st[0].ms_total for st in infos])),
                     "unaligned_device_ms": float(np.mean([st[1].ms_total for st in infos])) if n_un else None,
                     "aligned_kernel_ms": {nm: float(np.mean([st[0].ms_kernel[k] for st in infos])) for k, nm in enumerate(w.engine.KERNEL_NAMES)},
                     "unaligned_kernel_ms": {nm: float(np.mean([st[1].ms_kernel[k] for st in infos])) for k, nm in enumerate(w.engine.KERNEL_NAMES)} if n_un else None,
                     "note": "one engine context, aligned then unaligned worker call (CLI: NS_SERIAL=1): each call's kernels run ALONE here"}
    infos, dt, _ = timed_steps(w, a, n, n_al, n_un, steps, 2, None, True, first_step=3000)    # (two warm-ups: both error-profile slots sized)
    al = [st[0] for st in infos]
    k_err = w.engine.KERNEL_NAMES.index("k_errlog")
    ms_err = float(np.mean([x.ms_kernel[k_err] for x in al]))
    eb = float(np.mean([int(x.errlog_bytes) for x in al]))
    out["errlog_on"] = {"value": n * steps / dt, "unit": "reads/s", "ms_per_step": dt / steps * 1e3, "steps": steps,
                        "aligned_device_ms": float(np.mean([x.ms_total for x in al])),
                        "unaligned_device_ms": float(np.mean([st[1].ms_total for st in infos])) if n_un else None,
                        "k_errlog_ms": ms_err, "errlog_bytes_per_read": eb / max(1, n_al),
                        "k_errlog_store_gb_per_s": eb / (ms_err * 1e-3) / 1e9 if ms_err > 0 else None,
                        "k_errlog_frac": eb / (ms_err * 1e-3) / 1e9 / HBM_PEAK_GBS if ms_err > 0 else None,
                        "note": "two engine contexts as the headline, emit_errlog = 1 on the aligned worker call (k_errlen + k_errlog); "
                                "k_errlog_frac = error-profile bytes stored / k_errlog time / 8 TB/s"}
    return out


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
    ap.add_argument("--genome", choices=("ecoli", "chr1", "grch38"), default="ecoli",
                    help="ecoli: 4.64 Mb circular (configs[1], the default and the headline); chr1: 248.96 Mb linear (configs[2], with --fastq "
                         "--kmer-bias 5); grch38: 24 chromosomes, 3.1 Gb (configs[3], with --chimeric)")
    ap.add_argument("--chimeric", action="store_true", help="chimeric reads (S:1276-1299; configs[3] and, optionally, configs[4])")
    ap.add_argument("--metagenome", action="store_true", help="configs[4]: zymo10-like community, one metagenome worker call per step (S:814-1040)")
    ap.add_argument("--errlog", action="store_true", help="also format the error profile on the device")
    ap.add_argument("--trained-shape", action="store_true", help="the hg002_like rates in the table shape of a model read_analysis.py trains (15 previous-match bins, "
                                                                 "1 500-row ECDFs): the chain reads its tables from global memory")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-genome-run", action="store_true", help="same as --aligned-only (kept for the profiling scripts)")
    ap.add_argument("--cpu-sample", type=int, default=5000, help="reads PER CORE of the CPU baseline sample (about 10 s with every core busy)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) or gloo (single-GPU test of the N>1 path)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end legs (generation + D2H + file writes)")
    ap.add_argument("--e2e-steps", type=int, default=3)
    ap.add_argument("--e2e-dir", default="/dev/shm")
    ap.add_argument("--no-configs2", action="store_true", help="skip the configs[2] object (chr1-size reference, FASTQ, -hp -k 5)")
    ap.add_argument("--configs2-steps", type=int, default=3)
    ap.add_argument("--no-extras", action="store_true", help="skip the `serial` and `errlog_on` objects")
    ap.add_argument("--python-threads", action="store_true", help="the round-2..4 schedule: a second Engine with its own copies and a Python thread per "
                                                                  "step for the unaligned worker call, instead of ns_generate_step (A/B runs)")
    ap.add_argument("--no-solo-reference", action="store_true", help="N > 1: skip rank 0's single-rank repeat of the timed steps (multi_gpu.single_rank_reference)")
    ap.add_argument("--extras-steps", type=int, default=3)
    a = ap.parse_args()
    a.aligned_only = a.aligned_only or a.no_genome_run
    genome = "zymo10" if a.metagenome else a.genome
    default_cfg = genome == "ecoli" and not a.fastq and not a.kmer_bias and not a.aligned_only and not a.serial and not a.chimeric and not a.trained_shape
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(a)                                         # (does not return)

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
        # an N-GPU line is a line of N ranks over RCCL: anything else (a launcher that started fewer ranks, a build of torch that fell
        # back to another backend) must not pass as one
        if dist.get_world_size() != a.gpus:
            sys.exit("bench.py --gpus %d was launched with %d ranks" % (a.gpus, dist.get_world_size()))
        if a.dist_backend == "nccl" and dist.get_backend() != "nccl":
            sys.exit("bench.py --gpus %d: the process group reports backend %r, not nccl (= RCCL)" % (a.gpus, dist.get_backend()))
    torch.cuda.set_device(local_rank)

    # ---- inputs: synthetic hg002-like model (every rank, identical by seed) + the synthetic reference (rank 0) ----
    tmp = tempfile.mkdtemp(prefix="nsbench_%d_" % rank)
    w = Workload(a, genome, a.fastq, a.kmer_bias, local_rank, rank, world, dist, a.serial, a.aligned_only, tmp, chimeric=a.chimeric)
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

    solo = None
    if dist is not None and not a.no_solo_reference:
        # the same binary, the same workload, ONE rank busy: rank 0 repeats the timed steps while the others wait at the barrier — what the
        # N-rank value is an efficiency of (weak scaling: value / (N x this))
        if rank == 0:
            _, dt1, _ = timed_steps(w, a, n, n_al, n_un, a.steps, 1, None, a.errlog, first_step=5000)
            solo = {"value": n * a.steps / dt1, "unit": "reads/s", "ms_per_step": dt1 / a.steps * 1e3, "steps": a.steps,
                    "note": "rank 0 alone on its GPU, the other ranks idle at a barrier: same process, same engines, same reference copy"}
        dist.barrier()
    if rank == 0:
        out = summarise(w, a, infos, dt, n, n_al, n_un, a.steps, a.warmup, world, tot_bases, a.errlog)
        if world > 1:
            out["multi_gpu"] = {"world_size": dist.get_world_size(), "backend": "RCCL (torch.distributed nccl)" if a.dist_backend == "nccl" else a.dist_backend,
                                "backend_reported": dist.get_backend(),
                                "reference_broadcast_ms": w.broadcast_ms, "reference_bytes": w.glen,
                                "reference_broadcast_gb_per_s": w.glen / (w.broadcast_ms * 1e-3) / 1e9 if w.broadcast_ms else None,
                                "per_rank": per_rank, "collectives_in_timed_region": 0,
                                "single_rank_reference": solo,
                                "scaling_efficiency": out["value"] / (world * solo["value"]) if solo else None,
                                "ranks_on_one_gpu": os.environ.get("NS_BENCH_DEVICE") is not None}
        if world == 1 and not a.no_extras and w.eng_un is not None and n_un:
            try:
                out.update(extra_legs(w, a, n, n_al, n_un, a.extras_steps))
            except Exception as ex:
                out["serial"] = {"error": repr(ex)}
        if world == 1 and not a.no_e2e and w.eng_un is not None and not w.meta:
            try:
                out["e2e"] = {"d2h_pinned": d2h_rate(w), **e2e_legs(w, n, a.e2e_steps, a.e2e_dir)}
            except Exception as ex:                 # the headline must not depend on the state of /dev/shm
                out["e2e"] = {"error": repr(ex)}
        if not a.no_cpu_baseline and world == 1 and not w.meta:
            out["cpu_baseline"] = cpu_baseline(w.mdl, w.ref_host, engine, a.cpu_sample, a.fastq, a.kmer_bias, w.chimeric)
    w.close()
    if rank == 0 and world == 1 and default_cfg and not a.no_configs2:
        # on the same GPU, same protocol: BASELINE configs[2] (chr1-size linear reference, FASTQ + base qualities + homopolymers), and the
        # headline's FASTA workload on that reference — 249 Mb do not fit the L2 / Infinity Cache the 4.6 Mb of configs[1] live in, so this
        # is the record kernel's roofline with its source bytes coming from HBM
        import copy
        a_tr = copy.copy(a); a_tr.trained_shape = True
        # ... and the headline workload with its chain tables in the SHAPE read_analysis.py gives a trained model (15 previous-match bins x
        # 1 500-row ECDFs: --trained-shape): the hot prefixes of the columns in LDS, the rest in global memory
        for key, gen, fq, km, aa in (("configs2", "chr1", True, 5, a), ("chr1_fasta", "chr1", False, 0, a), ("trained_shape", "ecoli", False, 0, a_tr)):
            try:
                w2 = Workload(aa, gen, fq, km, local_rank, rank, world, None, False, False, tmp)
                infos2, dt2, _ = timed_steps(w2, a, n, n_al, n_un, a.configs2_steps, 2, None, False)
                c2 = summarise(w2, a, infos2, dt2, n, n_al, n_un, a.configs2_steps, 2, 1, sum(int(x.total_bases) for st in infos2 for x in st), False)
                out[key] = {k: c2[k] for k in ("value", "unit", "bases_per_s", "steps", "warmup", "ms_per_step", "config", "device_ms_per_step",
                                               "aligned_batch", "roofline", "unaligned_batch")}
                w2.close()
            except Exception as ex:
                out[key] = {"error": repr(ex)}
    if rank == 0:
        print(json.dumps(out))
        sys.stdout.flush()
    shutil.rmtree(tmp, ignore_errors=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
