"""Multi-GPU plumbing: one process per GPU (torch.distributed; backend "nccl" == RCCL on ROCm, "gloo" in CPU tests).

The reference's only parallelism is a fork fan-out that splits the read count and concatenates the workers'
sub-files in worker order (src/simulator.py:1588-1639, 1642-1672).  Here: read-index ranges are partitioned
across ranks, the reference genome is broadcast ONCE from rank 0, and there is no further collective — a read is
a pure function of (seed, read index), so the result does not depend on the number of GPUs.
"""
from __future__ import annotations

import os

import numpy as np

from .model import Reference


def partition(n: int, world: int) -> list[tuple[int, int]]:
    """[start, end) of every rank: rank g owns reads [g*n//G, (g+1)*n//G)  (SURVEY.md §8e)."""
    return [(g * n // world, (g + 1) * n // world) for g in range(world)]


def env_rank_world() -> tuple[int, int, int]:
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def broadcast_reference(ref: Reference | None, dist, device=None):
    """Rank 0 holds `ref`; every rank returns (Reference metadata, bases tensor on `device`).

    One broadcast of the concatenated genome bytes (the only data-path collective of a run) plus one small
    object broadcast for the chromosome table.  With device=None the tensor stays on the CPU (gloo tests).
    """
    import torch
    rank = dist.get_rank()
    meta = [None]
    if rank == 0:
        meta = [dict(names=ref.names, chrom_off=ref.chrom_off.tolist(), circular=ref.circular.tolist())]
    dist.broadcast_object_list(meta, src=0)
    m = meta[0]
    n = int(m["chrom_off"][-1])
    dev = device if device is not None else "cpu"
    buf = torch.empty(n, dtype=torch.uint8, device=dev)
    if rank == 0:
        buf.copy_(torch.from_numpy(np.ascontiguousarray(ref.bases)))
    dist.broadcast(buf, src=0)
    out = Reference(list(m["names"]), ref.bases if rank == 0 else np.zeros(0, np.uint8),
                    np.array(m["chrom_off"], dtype=np.uint64), np.array(m["circular"], dtype=np.uint8))
    return out, buf


def merge_subfiles(out_path: str, sub_paths: list[str], header: bytes = b"") -> None:
    """Concatenate the per-rank sub-files in rank order and remove them (S:1626-1639); a single sub-file is just renamed."""
    if len(sub_paths) == 1 and not header:
        os.replace(sub_paths[0], out_path)
        return
    with open(out_path, "wb") as out:
        if header:
            out.write(header)
        for p in sub_paths:
            with open(p, "rb") as f:
                while True:
                    chunk = f.read(1 << 24)
                    if not chunk:
                        break
                    out.write(chunk)
    for p in sub_paths:
        os.remove(p)
