"""Multi-GPU plumbing: one process per GPU (torch.distributed; backend "nccl" == RCCL on ROCm, "gloo" in CPU tests).

The reference's only parallelism is a fork fan-out that splits the read count and concatenates the workers'
sub-files in worker order (src/simulator.py:1588-1639, 1642-1672).  Here: read-index ranges are partitioned
across ranks, the reference genome is broadcast ONCE from rank 0, and there is no further data-path collective — a
read is a pure function of (seed, read index), so the result does not depend on the number of GPUs.  The ranks
write into the SAME output files at their final offsets (a sizing pass + one small all-gather of byte counts
tells every rank where its part starts), so there is no merge copy either.
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


def init_dist():
    """(rank, device index, world, dist module or None, torch device of the reference broadcast or None).

    One process per GPU under ``python -m torch.distributed.run``; the backend is RCCL ("nccl" on ROCm).  NS_DIST_BACKEND=gloo and
    NS_DEVICE=<index> exist for tests that run several ranks on ONE GPU: the reference then travels through host memory."""
    rank, local_rank, world = env_rank_world()
    device = int(os.environ.get("NS_DEVICE", local_rank))
    if world == 1:
        return rank, device, world, None, None
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    backend = os.environ.get("NS_DIST_BACKEND", "nccl")
    torch.cuda.set_device(device)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        return rank, device, world, dist, torch.device("cuda", device)
    dist.init_process_group(backend)
    return rank, device, world, dist, None


def agree(dist, ok: bool, message: str = "") -> None:
    """Every rank calls this before the next collective: if any rank failed its checks, ALL ranks exit with status 1 (a rank that
    exits alone leaves the others waiting in the collective until the RCCL timeout)."""
    if dist is None:
        if not ok:
            import sys
            sys.stderr.write(message)
            sys.exit(1)
        return
    import sys
    flags = [None] * dist.get_world_size()
    dist.all_gather_object(flags, (bool(ok), message))
    bad = [m for o, m in flags if not o]
    if bad:
        if dist.get_rank() == 0:
            sys.stderr.write(bad[0])
        dist.destroy_process_group()
        sys.exit(1)


def share_seed(dist, seed):
    """rank 0's seed for every rank: a read is a function of (seed, read index), so all ranks must draw from the same seed"""
    if dist is None:
        return seed
    box = [seed]
    dist.broadcast_object_list(box, src=0)
    return box[0]


def file_offsets(dist, sizes: tuple[int, ...]) -> tuple[tuple[int, ...], tuple[int, ...]]:
    """(first byte of this rank's part, total size) per file, from every rank's part sizes: ranks write in rank order (S:1626-1639)"""
    if dist is None:
        return tuple(0 for _ in sizes), tuple(sizes)
    all_sizes = [None] * dist.get_world_size()
    dist.all_gather_object(all_sizes, tuple(int(x) for x in sizes))
    r = dist.get_rank()
    return (tuple(sum(a[k] for a in all_sizes[:r]) for k in range(len(sizes))),
            tuple(sum(a[k] for a in all_sizes) for k in range(len(sizes))))


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
