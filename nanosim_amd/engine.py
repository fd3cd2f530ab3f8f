"""ctypes binding of the C-ABI (include/nanosim_amd.h) — the only way the Python host reaches the GPU.

There is NO CPU fallback: if libnanosim_amd.so is missing or no MI355X is visible, construction fails.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .model import (EVENT_DTYPE, PIECE_DTYPE, READ_DTYPE, Model, NsBatchInfo, NsModelTables, NsParams, Reference)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NANOSIM_AMD_LIB") or os.path.join(_HERE, "libnanosim_amd.so")   # override: A/B builds only
NS_BUF_RECORDS, NS_BUF_READS, NS_BUF_PIECES, NS_BUF_EVENTS, NS_BUF_ERRLOG, NS_BUF_POLYA, NS_BUF_SPLICED = 0, 1, 2, 3, 4, 5, 6
NS_SPLICED_BASE = 1 << 56
NS_EINVAL, NS_ENODEV, NS_ENOMEM, NS_EHIP, NS_ESTATE, NS_EIO = -1, -2, -3, -4, -5, -6
NS_KIND_ALIGNED, NS_KIND_UNALIGNED, NS_KIND_PERFECT = 0, 1, 2
NS_EMIT_SIZES = 2
KERNEL_NAMES = ("plan(k_nseg+k_lengths+scan+sort)", "k_chain", "k_names", "k_materialise", "k_hp", "k_errlog")
EXPORTS = ("ns_abi_version", "ns_create", "ns_destroy", "ns_last_error", "ns_set_reference",
           "ns_set_reference_device", "ns_load_model", "ns_generate", "ns_copy_out", "ns_device_ptr",
           "ns_set_species", "ns_set_abundance", "ns_species_bases", "ns_host_alloc", "ns_host_free",
           "ns_set_transcriptome", "ns_set_intron_retention", "ns_set_background",
           "ns_sink_open", "ns_sink_put", "ns_sink_write", "ns_sink_write_range", "ns_record_offsets", "ns_sink_drain",
           "ns_sink_close", "ns_io_counters", "ns_cs_histograms", "ns_generate_step", "ns_step_context", "ns_maf_histograms")


class NsIoStats(C.Structure):
    _fields_ = [("bytes", C.c_uint64), ("dma_ms", C.c_double), ("wait_staging_s", C.c_double), ("write_s", C.c_double),
                ("slice_bytes", C.c_uint64), ("n_slices", C.c_uint32), ("n_threads", C.c_uint32)]

_lib = None


class EngineError(RuntimeError):
    pass


def load_library(path: str = LIB_PATH):
    """dlopen the engine and declare every prototype of include/nanosim_amd.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise EngineError("%s not found — run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback" % path)
    L = C.CDLL(path)
    L.ns_abi_version.restype = C.c_uint32
    L.ns_create.restype = C.c_int
    L.ns_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.ns_destroy.restype = None
    L.ns_destroy.argtypes = [C.c_void_p]
    L.ns_last_error.restype = C.c_char_p
    L.ns_last_error.argtypes = [C.c_void_p]
    ref_args = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p, C.c_char_p, C.c_uint64]
    L.ns_set_reference.restype = C.c_int
    L.ns_set_reference.argtypes = ref_args
    L.ns_set_reference_device.restype = C.c_int
    L.ns_set_reference_device.argtypes = ref_args
    L.ns_load_model.restype = C.c_int
    L.ns_load_model.argtypes = [C.c_void_p, C.POINTER(NsModelTables)]
    L.ns_generate.restype = C.c_int
    L.ns_generate.argtypes = [C.c_void_p, C.POINTER(NsParams), C.POINTER(NsBatchInfo)]
    L.ns_copy_out.restype = C.c_int
    L.ns_copy_out.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.c_uint64]
    L.ns_device_ptr.restype = C.c_void_p
    L.ns_device_ptr.argtypes = [C.c_void_p, C.c_int]
    L.ns_set_species.restype = C.c_int
    L.ns_set_species.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    L.ns_set_abundance.restype = C.c_int
    L.ns_set_abundance.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.ns_set_background.restype = C.c_int
    L.ns_set_background.argtypes = [C.c_void_p, C.c_int]
    L.ns_species_bases.restype = C.c_int
    L.ns_species_bases.argtypes = [C.c_void_p, C.c_void_p]
    L.ns_set_transcriptome.restype = C.c_int
    L.ns_set_transcriptome.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double]
    L.ns_set_intron_retention.restype = C.c_int
    L.ns_set_intron_retention.argtypes = [C.c_void_p, C.c_void_p]
    L.ns_host_alloc.restype = C.c_int
    L.ns_host_alloc.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]
    L.ns_host_free.restype = C.c_int
    L.ns_host_free.argtypes = [C.c_void_p, C.c_void_p]
    L.ns_sink_open.restype = C.c_int
    L.ns_sink_open.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.POINTER(C.c_void_p)]
    L.ns_sink_put.restype = C.c_int
    L.ns_sink_put.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_uint64]
    L.ns_sink_write.restype = C.c_int
    L.ns_sink_write.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.ns_sink_write_range.restype = C.c_int
    L.ns_sink_write_range.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_uint64]
    L.ns_record_offsets.restype = C.c_int
    L.ns_record_offsets.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    L.ns_sink_drain.restype = C.c_int
    L.ns_sink_drain.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]
    L.ns_sink_close.restype = C.c_int
    L.ns_sink_close.argtypes = [C.c_void_p, C.c_void_p]
    L.ns_io_counters.restype = C.c_int
    L.ns_io_counters.argtypes = [C.c_void_p, C.POINTER(NsIoStats), C.c_int]
    L.ns_cs_histograms.restype = C.c_int
    L.ns_cs_histograms.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p]
    L.ns_maf_histograms.restype = C.c_int
    L.ns_maf_histograms.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p]
    L.ns_generate_step.restype = C.c_int
    L.ns_generate_step.argtypes = [C.c_void_p, C.POINTER(NsParams), C.POINTER(NsParams), C.POINTER(NsBatchInfo)]
    L.ns_step_context.restype = C.c_int
    L.ns_step_context.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    _lib = L
    return L


class Sink:
    """An output file the engine appends result buffers to (ns_sink_*): the out_reads / out_error handle of a worker
    (src/simulator.py:1437-1443, 2006-2008).  write() queues a buffer of the LAST batch and returns at once; the bytes leave the GPU
    on the engine's copy stream while the next batch is generated."""

    def __init__(self, eng: "Engine", fd: int, file_off: int = 0):
        self.eng = eng
        self.h = C.c_void_p()
        eng._check(eng.L.ns_sink_open(eng.ctx, fd, file_off, C.byref(self.h)))

    def put(self, data: bytes):
        self.eng._check(self.eng.L.ns_sink_put(self.eng.ctx, self.h, data, len(data)))

    def write(self, which: int, offset: int = 0, nbytes: int | None = None):
        """queue the whole buffer `which` of the last batch, or its bytes [offset, offset + nbytes)"""
        if offset == 0 and nbytes is None:
            self.eng._check(self.eng.L.ns_sink_write(self.eng.ctx, self.h, which))
        else:
            self.eng._check(self.eng.L.ns_sink_write_range(self.eng.ctx, self.h, which, offset, nbytes))

    def drain(self) -> int:
        """wait until everything queued is in the file; returns the file offset behind the last byte"""
        off = C.c_uint64()
        self.eng._check(self.eng.L.ns_sink_drain(self.eng.ctx, self.h, C.byref(off)))
        return int(off.value)

    def close(self):
        if self.h:
            h, self.h = self.h, C.c_void_p()
            self.eng._check(self.eng.L.ns_sink_close(self.eng.ctx, h))


class Batch:
    """Results of one ns_generate call; buffers stay in HBM until the next call on the same engine."""

    def __init__(self, eng: "Engine", info: NsBatchInfo):
        self.eng = eng
        self.info = info

    def _copy(self, which, dtype, count):
        out = np.empty(count, dtype=dtype)
        if count:
            self.eng._check(self.eng.L.ns_copy_out(self.eng.ctx, which, out.ctypes.data, 0, out.nbytes))
        return out

    def reads(self):
        return self._copy(NS_BUF_READS, READ_DTYPE, int(self.info.n_reads))

    def pieces(self):
        return self._copy(NS_BUF_PIECES, PIECE_DTYPE, int(self.info.n_pieces))

    def events(self):
        return self._copy(NS_BUF_EVENTS, EVENT_DTYPE, int(self.info.n_events))

    def records(self, out: np.ndarray | None = None):
        n = int(self.info.record_bytes)
        if out is None:
            return self._copy(NS_BUF_RECORDS, np.uint8, n)
        if n:
            self.eng._check(self.eng.L.ns_copy_out(self.eng.ctx, NS_BUF_RECORDS, out.ctypes.data, 0, n))
        return out[:n]

    def errlog(self):
        return self._copy(NS_BUF_ERRLOG, np.uint8, int(self.info.errlog_bytes))

    def polya(self):
        """polyA tail length per read (transcriptome batches)"""
        return self._copy(NS_BUF_POLYA, np.uint16, int(self.info.n_reads))

    def spliced(self):
        """intron retention: the splice arena of the batch (pieces with ref_gpos >= NS_SPLICED_BASE point into it)"""
        return self._copy(NS_BUF_SPLICED, np.uint8, int(self.info.spliced_bytes))

    def record_offsets(self, read_index):
        """(record offsets, error-profile offsets) of the given read indices of this batch (n_reads = the end of the images)"""
        idx = np.ascontiguousarray(read_index, dtype=np.uint64)
        rec, err = np.zeros(len(idx), dtype=np.uint64), np.zeros(len(idx), dtype=np.uint64)
        self.eng._check(self.eng.L.ns_record_offsets(self.eng.ctx, idx.ctypes.data, len(idx), rec.ctypes.data, err.ctypes.data))
        return rec, err

    def kernel_ms(self):
        return {KERNEL_NAMES[i]: float(self.info.ms_kernel[i]) for i in range(len(KERNEL_NAMES))}

    def copy_range(self, which: int, offset: int, out: np.ndarray, nbytes: int):
        """bytes [offset, offset + nbytes) of a result buffer into `out` (ideally from Engine.pinned)"""
        if nbytes:
            self.eng._check(self.eng.L.ns_copy_out(self.eng.ctx, which, out.ctypes.data, offset, nbytes))
        return out[:nbytes]


class Engine:
    """One context per GPU (single host thread), mirroring the worker seam of src/simulator.py:1601-1619."""

    def __init__(self, device: int = 0):
        self.L = load_library()
        self.ctx = C.c_void_p()
        rc = self.L.ns_create(device, C.byref(self.ctx))
        if rc != 0:
            raise EngineError("ns_create(device=%d) failed with %d (no MI355X visible?)" % (device, rc))
        self._keep = []
        self._pinned = []

    @classmethod
    def _borrowed(cls, owner: "Engine", ctx: C.c_void_p) -> "Engine":
        """an Engine over a context the library owns (the step companion of `owner`): every result call works on it, close() leaves it alone"""
        e = cls.__new__(cls)
        e.L, e.ctx, e._keep, e._pinned, e._owner = owner.L, ctx, [], [], owner
        return e

    def step_engine(self) -> "Engine":
        """The context the UNALIGNED worker call of generate_step runs on (ns_step_context): it shares this engine's reference, model and
        mode tables; its batch buffers, sinks and I/O counters are its own.  Owned by this engine (closed with it)."""
        if getattr(self, "_step_engine", None) is None:
            c = C.c_void_p()
            self._check(self.L.ns_step_context(self.ctx, C.byref(c)))
            self._step_engine = Engine._borrowed(self, c)
        return self._step_engine

    def generate_step(self, aligned: NsParams | None, unaligned: NsParams | None):
        """One step of simulation() (S:1571-1672): the aligned and the unaligned worker call side by side on this GPU (ns_generate_step).
        Returns (aligned Batch or None, unaligned Batch or None); the unaligned batch lives on step_engine()."""
        info = (NsBatchInfo * 2)()
        un_eng = self.step_engine() if unaligned is not None else None
        try:
            self._check(self.L.ns_generate_step(self.ctx, C.byref(aligned) if aligned is not None else None,
                                                C.byref(unaligned) if unaligned is not None else None, info))
        except EngineError as err:
            # only the unaligned half failed (the library says which: "unaligned worker call: ..."): the aligned batch is complete
            if aligned is not None and unaligned is not None and "unaligned worker call:" in str(err):
                err.aligned_batch = Batch(self, info[0])
            raise
        return (Batch(self, info[0]) if aligned is not None else None, Batch(un_eng, info[1]) if unaligned is not None else None)

    def set_background(self, on: bool = True):
        """this context's worker calls run next to another context's on the same GPU (ns_set_background)"""
        self._check(self.L.ns_set_background(self.ctx, 1 if on else 0))

    def close(self):
        if self.ctx:
            for p in self._pinned:
                self.L.ns_host_free(self.ctx, p)
            self._pinned = []
            if getattr(self, "_owner", None) is not None:      # a step companion: its owner destroys it
                self.ctx = C.c_void_p()
                if getattr(self._owner, "_step_engine", None) is self:      # ... and hands out a fresh wrapper on the next step_engine()
                    self._owner._step_engine = None
                return
            se = getattr(self, "_step_engine", None)
            if se is not None:
                se.close()
                self._step_engine = None
            self.L.ns_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            err = EngineError("nanosim_amd error %d: %s" % (rc, self.L.ns_last_error(self.ctx).decode()))
            err.code = rc
            raise err

    def set_reference(self, ref: Reference):
        blob = ref.names_blob()
        self._check(self.L.ns_set_reference(self.ctx, ref.bases.ctypes.data, len(ref.bases), ref.chrom_off.ctypes.data,
                                            len(ref.names), ref.circular.ctypes.data, blob, len(blob)))

    def set_reference_device(self, dev_ptr: int, ref: Reference):
        """`dev_ptr` points at ref.genome_len bytes already resident on this GPU (e.g. after an RCCL
        broadcast through torch.distributed); the engine copies them device-to-device (the caller may free its buffer)."""
        blob = ref.names_blob()
        self._check(self.L.ns_set_reference_device(self.ctx, dev_ptr, ref.genome_len, ref.chrom_off.ctypes.data,
                                                   len(ref.names), ref.circular.ctypes.data, blob, len(blob)))

    def set_metagenome(self, meta_ref, abun: dict | None = None, abun_inflated: dict | None = None, dev_ptr: int | None = None):
        """meta_ref: nanosim_amd.metagenome.MetaReference; abun/abun_inflated: {species: value} of the sample (may be
        set later, per sample, with set_abundance); dev_ptr: the concatenated bases already on this GPU."""
        if dev_ptr is None:
            self.set_reference(meta_ref.ref)
        else:
            self.set_reference_device(dev_ptr, meta_ref.ref)
        sco = np.ascontiguousarray(meta_ref.species_chrom_off, dtype=np.uint32)
        self._check(self.L.ns_set_species(self.ctx, len(meta_ref.species), sco.ctypes.data))
        self._nspecies = len(meta_ref.species)
        if abun is not None:
            self.set_abundance(meta_ref, abun, abun_inflated)

    def set_transcriptome(self, tr, dev_ptr: int | None = None):
        """tr: nanosim_amd.transcriptome.TranscriptomeReference (transcripts = chromosomes, expression weights, polyA list)"""
        if dev_ptr is None:
            self.set_reference(tr.ref)
        else:
            self.set_reference_device(dev_ptr, tr.ref)
        ec = np.ascontiguousarray(tr.expr_chrom, dtype=np.uint32)
        cum = np.ascontiguousarray(tr.expr_cum, dtype=np.float64)
        pa = np.ascontiguousarray(tr.polya, dtype=np.uint8)
        self._check(self.L.ns_set_transcriptome(self.ctx, len(ec), ec.ctypes.data, cum.ctypes.data, pa.ctypes.data if pa.any() else None,
                                                float(tr.polya_scale)))

    def set_intron_retention(self, ir):
        """ir: nanosim_amd.intron_retention.IntronRetention for the transcriptome set before (None: off)"""
        if ir is None:
            self._check(self.L.ns_set_intron_retention(self.ctx, None))
            return
        t = ir.to_c()
        self._check(self.L.ns_set_intron_retention(self.ctx, C.byref(t)))

    def set_abundance(self, meta_ref, abun: dict, abun_inflated: dict | None = None):
        # a genome of the list that the abundance table does not name gets no quota (its chromosomes still serve gaps and unaligned
        # reads, as in the reference, whose quotas run over dict_abun only, S:772-775)
        ab = np.array([abun.get(sp, 0.0) for sp in meta_ref.species], dtype=np.float64)
        inf = np.array([abun_inflated.get(sp, 0.0) for sp in meta_ref.species], dtype=np.float64) if abun_inflated else None
        self._check(self.L.ns_set_abundance(self.ctx, ab.ctypes.data, inf.ctypes.data if inf is not None else None))
        self._nspecies = len(meta_ref.species)

    def species_bases(self) -> np.ndarray:
        out = np.zeros(self._nspecies, dtype=np.float64)
        self._check(self.L.ns_species_bases(self.ctx, out.ctypes.data))
        return out

    def pinned(self, nbytes: int) -> np.ndarray:
        """uint8 array over page-locked host memory (ns_host_alloc); freed with the engine"""
        p = C.c_void_p()
        self._check(self.L.ns_host_alloc(self.ctx, nbytes, C.byref(p)))
        self._pinned.append(p)
        return np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(p.value))

    def sink(self, fd: int, file_off: int = 0) -> Sink:
        return Sink(self, fd, file_off)

    def io_counters(self, reset: bool = False) -> dict:
        st = NsIoStats()
        self._check(self.L.ns_io_counters(self.ctx, C.byref(st), 1 if reset else 0))
        return dict(bytes=int(st.bytes), dma_ms=float(st.dma_ms), wait_staging_s=float(st.wait_staging_s), write_s=float(st.write_s),
                    d2h_gbs=(st.bytes / (st.dma_ms * 1e-3) / 1e9) if st.dma_ms > 0 else None,
                    slice_bytes=int(st.slice_bytes), n_slices=int(st.n_slices), n_threads=int(st.n_threads))

    def load_model(self, model: Model):
        t = model.to_c()
        self._check(self.L.ns_load_model(self.ctx, C.byref(t)))

    def generate(self, params: NsParams) -> Batch:
        info = NsBatchInfo()
        self._check(self.L.ns_generate(self.ctx, C.byref(params), C.byref(info)))
        return Batch(self, info)


def make_params(*, seed, first_read, n_reads, kind=NS_KIND_ALIGNED, fastq=False, kmer_bias=0, chimeric=False,
                min_len=50, max_len, median_len=None, sd_len=None, emit_records=True, emit_errlog=False, meta=False, trx=False, uracil=False,
                model_ir=False) -> NsParams:
    p = NsParams()
    p.seed, p.first_read, p.n_reads, p.kind = seed, first_read, n_reads, kind
    p.fastq, p.kmer_bias, p.chimeric = int(bool(fastq)), int(kmer_bias or 0), int(bool(chimeric))
    p.use_lognormal = int(median_len is not None and sd_len is not None)
    p.emit_records = NS_EMIT_SIZES if emit_records == "sizes" else int(bool(emit_records))     # "sizes": record_bytes / errlog_bytes only
    p.emit_errlog = int(bool(emit_errlog))
    p.min_len, p.max_len = int(min_len), int(max_len)
    p.median_len, p.sd_len = float(median_len or 0.0), float(sd_len or 0.0)
    p.meta = int(bool(meta))
    p.trx = int(bool(trx))
    p.uracil = int(bool(uracil))
    p.model_ir = int(bool(model_ir))
    return p
