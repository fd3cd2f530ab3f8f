"""Multi-GPU plumbing: one process per GPU (torch.distributed; backend "nccl" == RCCL on ROCm, "gloo" in CPU tests).

The reference's only parallelism is a fork fan-out that splits the read count and concatenates the workers'
sub-files in worker order (src/simulator.py:1588-1639, 1642-1672).  Here: read-index ranges are partitioned
across ranks, and the reference genome is broadcast ONCE from rank 0 — the run's single collective: the
chromosome table, the seed and the mode's small tables ride behind the bases in the same buffer, and the size of that
buffer is published through the rendezvous key-value store, which is not a collective.  A read is a pure function of
(seed, read index), so the result does not depend on the number of GPUs.  Output follows the reference (S:1626-1639):
rank 0 writes the head of every output file, every other rank a sub-file `<file>.part<rank>`, and rank 0 appends the
sub-files in rank order as they appear.  Completion and failure travel through the file system (a finished sub-file is
renamed into place, a failed rank leaves `<file>.part<rank>.failed`), so no rank ever waits inside a collective for a
peer that has died.
"""
from __future__ import annotations

import os
import pickle
import sys
import time

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
    if world == 1 and os.environ.get("NS_FORCE_DIST", "0") == "0":      # (NS_FORCE_DIST=1: a process group of one rank — the RCCL path on a 1-GPU box)
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
    exits alone leaves the others waiting in the collective until the RCCL timeout).  The CLI no longer needs it — rank 0's checks
    travel in the header of the reference broadcast — it stays for callers that validate on every rank."""
    if dist is None:
        if not ok:
            sys.stderr.write(message)
            sys.exit(1)
        return
    flags = [None] * dist.get_world_size()
    dist.all_gather_object(flags, (bool(ok), message))
    bad = [m for o, m in flags if not o]
    if bad:
        if dist.get_rank() == 0:
            sys.stderr.write(bad[0])
        dist.destroy_process_group()
        sys.exit(1)


def _default_store():
    """the rendezvous store of the default process group, or None when this torch has no such accessor.  Whether it exists depends on the
    torch build and on how the group was created — the same on every rank — so every rank takes the same transport below."""
    try:
        from torch.distributed import distributed_c10d as c10d
        return c10d._get_default_store()
    except (ImportError, AttributeError):
        return None


def _publish_header(dist, key: str, payload: bytes | None) -> bytes:
    """rank 0's `payload` for every rank through the rendezvous store of the process group (a TCP key-value store: set / blocking get,
    no collective).  Only a torch WITHOUT the store accessor takes the fallback, an object broadcast — decided before any rank
    publishes, identically on all ranks; an error of set / get itself propagates (a rank that fell back on its own would sit in a
    collective its peers never enter)."""
    store = _default_store()
    if store is None:
        box = [payload]
        dist.broadcast_object_list(box, src=0)
        return box[0]
    if dist.get_rank() == 0:
        store.set(key, payload)
        return payload
    return bytes(store.get(key))


_BCAST_SEQ = [0]


def broadcast_reference(ref: Reference | None, dist, device=None, extra=None, error: str | None = None):
    """Rank 0 holds `ref` (and `extra`, any picklable control data: the seed, the species / expression tables ...); every rank
    returns (Reference metadata, bases tensor on `device`, extra).

    ONE broadcast: the buffer is [bases | pickle(names, chrom_off, circular, extra)]; its two lengths reach the other ranks
    through the rendezvous store.  `error` (rank 0): a failed check — every rank prints nothing, rank 0 the message, and ALL ranks
    exit with status 1 together (S:354-356 & co: the reference exits before it forks).  device=None: the tensor stays on the CPU (gloo)."""
    import torch
    rank = dist.get_rank()
    _BCAST_SEQ[0] += 1
    key = "ns_ref_header_%d" % _BCAST_SEQ[0]
    blob = b""
    if rank == 0:
        if error is None:
            blob = pickle.dumps(dict(names=ref.names, chrom_off=np.asarray(ref.chrom_off), circular=np.asarray(ref.circular), extra=extra),
                                protocol=pickle.HIGHEST_PROTOCOL)
        hdr = pickle.dumps(dict(n=0 if error else int(ref.chrom_off[-1]), m=len(blob), error=error))
    hdr = pickle.loads(_publish_header(dist, key, hdr if rank == 0 else None))
    if hdr["error"] is not None:
        if rank == 0:
            sys.stderr.write(hdr["error"])
        dist.destroy_process_group()
        sys.exit(1)
    n, m = int(hdr["n"]), int(hdr["m"])
    dev = device if device is not None else "cpu"
    buf = torch.empty(n + m, dtype=torch.uint8, device=dev)
    if rank == 0:
        buf[:n].copy_(torch.from_numpy(np.ascontiguousarray(ref.bases)))
        buf[n:].copy_(torch.frombuffer(bytearray(blob), dtype=torch.uint8))
    dist.broadcast(buf, src=0)                                   # the run's one collective: RCCL over xGMI
    meta = pickle.loads(buf[n:].cpu().numpy().tobytes())
    out = Reference(list(meta["names"]), ref.bases if rank == 0 else np.zeros(0, np.uint8),
                    np.asarray(meta["chrom_off"], dtype=np.uint64), np.asarray(meta["circular"], dtype=np.uint8))
    return out, buf[:n], meta["extra"]


# ---- output files of a multi-rank run (S:1626-1639) ------------------------------------------------------------------------------

def part_path(path: str, rank: int) -> str:
    """the file rank `rank` writes: rank 0 the final file itself (its records are the head of it), the others a sub-file"""
    return path if rank == 0 else "%s.part%d" % (path, rank)


def subfile_path(path: str, tag) -> str:
    """sub-file `tag` of an output file, named as the reference names its workers' sub-files (S:1594-1595, 1650):
    <out>_aligned_reads<i>.fasta, <out>_error_profile<i> (without "aligned_"), <out>_unaligned_reads<i>.fasta"""
    head, ext = (path[:-6], path[-6:]) if path.endswith((".fasta", ".fastq")) else (path, "")
    if head.endswith("_aligned_error_profile"):
        head = head[:-len("_aligned_error_profile")] + "_error_profile"
    return "%s%s%s" % (head, tag, ext)


def clean_parts(paths, rank: int) -> None:
    """before the run's broadcast: no sub-file list, temporary or failure marker of an earlier run may be mistaken for this run's"""
    if rank == 0:
        return
    for p in paths:
        for q in (part_path(p, rank), part_path(p, rank) + ".subfiles", part_path(p, rank) + ".subfiles.tmp", part_path(p, rank) + ".failed"):
            try:
                os.unlink(q)
            except FileNotFoundError:
                pass


def publish_parts(path: str, rank: int, files) -> None:
    """rank > 0 is done with `path`: the names of the files that hold its bytes, in order, appear atomically as <path>.part<rank>.subfiles"""
    tmp = part_path(path, rank) + ".subfiles.tmp"
    with open(tmp, "w") as f:
        f.write("".join(os.path.abspath(x) + "\n" for x in files))
    os.rename(tmp, part_path(path, rank) + ".subfiles")


def mark_failed(path: str, rank: int, message: str) -> None:
    try:
        with open("%s.part%d.failed" % (path, rank), "w") as f:
            f.write(message + "\n")
    except OSError:
        pass


class failure_markers:
    """`with failure_markers(outputs, rank, world):` around everything a rank > 0 does after the broadcast: whatever ends the block with
    an exception — a model that does not load, an engine that cannot be created, ENOMEM in a later batch, a failed write — leaves
    `<file>.part<rank>.failed` for EVERY output, so rank 0 (collect_parts) ends with status 1 at once instead of polling for a sub-file
    list that never comes.  (A rank killed by a signal leaves nothing: the launcher tears the job down, and NS_PART_TIMEOUT bounds the wait.)"""

    def __init__(self, paths, rank: int, world: int):
        self.paths, self.rank, self.world = list(paths), rank, world

    def __enter__(self):
        return self

    def __exit__(self, et, ev, tb):
        if et is not None and self.rank > 0 and self.world > 1 and not (et is SystemExit and ev.code in (0, None)):
            for p in self.paths:
                mark_failed(p, self.rank, repr(ev))
        return False


def _append_file(dst_fd: int, src_path: str) -> int:
    """the bytes of src_path appended to dst_fd inside the kernel (copy_file_range: a reflink where the file system has one), falling
    back to sendfile / read + write"""
    total = 0
    with open(src_path, "rb") as src:
        size = os.fstat(src.fileno()).st_size
        mode = "cfr" if hasattr(os, "copy_file_range") else "sendfile"
        while total < size:
            want = min(1 << 30, size - total)
            try:
                if mode == "cfr":
                    n = os.copy_file_range(src.fileno(), dst_fd, want)
                elif mode == "sendfile":
                    n = os.sendfile(dst_fd, src.fileno(), None, want)
                else:
                    chunk = src.read(min(want, 64 << 20))
                    n = len(chunk)
                    view = memoryview(chunk)
                    while view:
                        w = os.write(dst_fd, view)
                        view = view[w:]
            except OSError:
                if mode == "rw":
                    raise
                mode = "sendfile" if mode == "cfr" else "rw"       # EXDEV / EINVAL / ENOSYS: next method, same position
                if mode == "rw":
                    src.seek(total)
                continue
            if n == 0:
                break
            total += n
    return total


def collect_parts(path: str, world: int, own=None, keep: bool = False, timeout_s: float | None = None, poll_s: float = 0.02) -> None:
    """Rank 0, once its own bytes of `path` are written (`own`: the files that hold them, in order — [path] itself when it wrote the final
    file directly): append everything else in order — its own sub-files, then those of ranks 1 .. world-1 as their lists appear
    (publish_parts) — and remove the sub-files.  keep: nothing is copied; <path>.subfiles lists the files in order instead.
    A `.failed` marker of any rank (or the timeout, NS_PART_TIMEOUT seconds, default one hour — counted from the moment rank 0 is
    done with its own share: the ranks have equal shares and finish together) ends the run with status 1."""
    if timeout_s is None:
        timeout_s = float(os.environ.get("NS_PART_TIMEOUT", "3600"))
    deadline = time.monotonic() + timeout_s
    own = [os.path.abspath(x) for x in (own if own is not None else [path])]
    final = os.path.abspath(path)
    order = list(own)
    fd = None
    try:
        if not keep:
            if world > 1:            # (--merge with several ranks: say what it will cost before it does)
                try:
                    own_bytes = sum(os.path.getsize(x) for x in own)
                except OSError:
                    own_bytes = 0
                sys.stderr.write("merging %d ranks' parts of %s into one file: about %.1f GB through one inode at 6-10 GB/s = %.0f-%.0f s "
                                 "(default without --merge: the parts stay, <file>.subfiles lists them)\n"
                                 % (world, os.path.basename(path), own_bytes * world / 1e9, own_bytes * (world - 1) / 10e9, own_bytes * (world - 1) / 6e9))
            fd = os.open(path, os.O_WRONLY | os.O_CREAT, 0o644)          # (no O_APPEND: copy_file_range refuses such a descriptor)
            if final not in own:
                os.ftruncate(fd, 0)
            os.lseek(fd, 0, os.SEEK_END)
            for x in own:
                if x != final:
                    _append_file(fd, x)
                    os.unlink(x)
        for r in range(1, world):
            lst = part_path(path, r) + ".subfiles"
            while not os.path.exists(lst):
                bad = [q for q in (part_path(path, k) + ".failed" for k in range(1, world)) if os.path.exists(q)]
                if bad or time.monotonic() > deadline:
                    msg = open(bad[0]).read().strip() if bad else "timed out waiting for " + lst
                    sys.stderr.write("\nrank failure while writing %s: %s\n" % (path, msg))
                    sys.exit(1)
                time.sleep(poll_s)
            names = [x for x in open(lst).read().split("\n") if x]
            for x in names:
                if keep:
                    order.append(x)
                else:
                    _append_file(fd, x)
                    os.unlink(x)
            os.unlink(lst)
        if keep and order != [final]:
            with open(path + ".subfiles", "w") as f:
                f.write("".join(x + "\n" for x in order))
    finally:
        if fd is not None:
            os.close(fd)
