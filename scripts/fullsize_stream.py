#!/usr/bin/env python3
"""BASELINE.json's configurations at their FULL read counts, through a size-independent property: the record stream of ALL N reads of a
configuration is generated twice under different partitions of the read indices —
  A: one worker, batches of --batch-a reads;
  B: --ranks contiguous index ranges (what --ranks GPUs would take, nanosim_amd/shard.py), each in batches of --batch-b reads —
and must be the same bytes: same XXH3-64 of the stream in read-index order, same byte / read / base totals.  (A read is a function of
(seed, read index); the GPU parity tests hold that against the oracle at 10^4..2x10^6 reads, this run holds the batching and sharding
independence at the 10^7 / 10^8 reads the configurations name.)  The bytes cross PCIe into page-locked memory and are hashed on the host.

    python scripts/fullsize_stream.py --config 2          # chr1_like, FASTQ, -hp -k 5, 10^7 reads
    python scripts/fullsize_stream.py --config 3          # grch38_like, --chimeric, 10^8 reads
"""
import argparse
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import xxhash

import bench

CONFIGS = {1: dict(genome="ecoli", fastq=False, kmer=0, chimeric=False, reads=10**6),
           2: dict(genome="chr1", fastq=True, kmer=5, chimeric=False, reads=10**7),
           3: dict(genome="grch38", fastq=False, kmer=0, chimeric=True, reads=10**8)}


def ranges(lo, hi, parts):
    """the contiguous index ranges of [lo, hi) that `parts` GPUs take (nanosim_amd/shard.py: partition)"""
    from nanosim_amd import shard
    return [(lo + a, lo + b) for a, b in shard.partition(hi - lo, parts)]


def stream(w, kind, spans, batch, pin):
    """XXH3-64 and totals of the records of the reads of `spans` (in order), generated `batch` reads per worker call"""
    E = w.engine
    h = xxhash.xxh3_64()
    tot = dict(reads=0, bytes=0, bases=0, calls=0)
    for lo, hi in spans:
        for first in range(lo, hi, batch):
            n = min(batch, hi - first)
            p = E.make_params(seed=bench.SEED, first_read=first, n_reads=n, kind=kind, fastq=w.fastq, max_len=w.max_len, kmer_bias=w.kmer if kind == E.NS_KIND_ALIGNED else 0,
                              chimeric=w.chimeric and kind == E.NS_KIND_ALIGNED)
            b = w.eng.generate(p)
            nb = int(b.info.record_bytes)
            rec = b.records(out=pin) if nb <= len(pin) else b.records()
            h.update(rec)
            tot["reads"] += int(b.info.n_reads); tot["bytes"] += nb; tot["bases"] += int(b.info.total_bases); tot["calls"] += 1
    tot["xxh3_64"] = h.hexdigest()
    return tot


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--reads", type=int, default=None, help="default: the configuration's N")
    ap.add_argument("--batch-a", type=int, default=500000)
    ap.add_argument("--batch-b", type=int, default=333333)
    ap.add_argument("--ranks", type=int, default=8)
    a = ap.parse_args()
    c = CONFIGS[a.config]
    n = a.reads or c["reads"]
    t0 = time.time()
    with tempfile.TemporaryDirectory() as tmp:
        ns = argparse.Namespace(python_threads=False, dist_backend="nccl")
        w = bench.Workload(ns, c["genome"], c["fastq"], c["kmer"], 0, 0, 1, None, True, False, tmp, chimeric=c["chimeric"])
        E = w.engine
        n_al, n_un = w.mdl.split_counts(n)
        per_read = 2 * 12000 if c["fastq"] else 12000
        pin = w.eng.pinned(max(a.batch_a, a.batch_b) * per_read)
        out = dict(config=w.describe(), reads=n, aligned=n_al, unaligned=n_un, seed=bench.SEED, setup_s=round(time.time() - t0, 1))
        for kind, name, lo, hi in ((E.NS_KIND_ALIGNED, "aligned", 0, n_al), (E.NS_KIND_UNALIGNED, "unaligned", n_al, n)):
            t1 = time.time()
            A = stream(w, kind, [(lo, hi)], a.batch_a, pin)
            t2 = time.time()
            B = stream(w, kind, ranges(lo, hi, a.ranks), a.batch_b, pin)
            t3 = time.time()
            same = all(A[k] == B[k] for k in ("reads", "bytes", "bases", "xxh3_64"))
            out[name] = dict(one_stream=dict(A, batch=a.batch_a, seconds=round(t2 - t1, 1)), ranks=dict(B, ranks=a.ranks, batch=a.batch_b, seconds=round(t3 - t2, 1)), identical=same)
            print(json.dumps({name: out[name]}), flush=True)
        out["identical"] = bool(out["aligned"]["identical"] and out["unaligned"]["identical"])
        w.close()
    print(json.dumps(out))
    return 0 if out["identical"] else 1


if __name__ == "__main__":
    sys.exit(main())
