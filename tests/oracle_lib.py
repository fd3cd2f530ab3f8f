"""ctypes binding of oracle/libns_oracle.so — TEST INFRASTRUCTURE ONLY (never imported by nanosim_amd)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from nanosim_amd.model import (EVENT_DTYPE, PIECE_DTYPE, READ_DTYPE, NsModelTables, NsParams)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None


class NsoDraw(C.Structure):
    _fields_ = [("mode", C.c_int), ("seed", C.c_uint64), ("read", C.c_uint64),
                ("tape_u", C.POINTER(C.c_double)), ("n_u", C.c_uint64), ("i_u", C.c_uint64),
                ("tape_n", C.POINTER(C.c_int64)), ("n_n", C.c_uint64), ("i_n", C.c_uint64),
                ("tape_z", C.POINTER(C.c_double)), ("n_z", C.c_uint64), ("i_z", C.c_uint64),
                ("tape_err", C.c_int)]


class NsoElist(C.Structure):
    _fields_ = [("l_new", C.c_int64), ("middle_ref", C.c_int64), ("e_count", C.c_int64 * 3),
                ("n_ev", C.c_uint64), ("shift", C.c_int64), ("overflow", C.c_int), ("range", C.c_int)]


class NsoOut(C.Structure):
    _fields_ = [("reads", C.c_void_p), ("pieces", C.c_void_p), ("events", C.c_void_p),
                ("cap_pieces", C.c_uint64), ("cap_events", C.c_uint64),
                ("records", C.c_void_p), ("cap_records", C.c_uint64),
                ("errlog", C.c_void_p), ("cap_errlog", C.c_uint64),
                ("n_pieces", C.c_uint64), ("n_events", C.c_uint64), ("record_bytes", C.c_uint64),
                ("errlog_bytes", C.c_uint64), ("total_bases", C.c_uint64), ("total_ref_bases", C.c_uint64),
                ("polya", C.c_void_p), ("spliced", C.c_void_p), ("cap_spliced", C.c_uint64), ("spliced_bytes", C.c_uint64)]


class NsoTrx(C.Structure):
    _fields_ = [("n_expr", C.c_uint32), ("expr_chrom", C.POINTER(C.c_uint32)), ("expr_cum", C.POINTER(C.c_double)),
                ("polya", C.POINTER(C.c_uint8)), ("polya_scale", C.c_double), ("ir", C.c_void_p)]


class NsoMeta(C.Structure):
    _fields_ = [("nspecies", C.c_uint32), ("species_chrom_off", C.POINTER(C.c_uint32)), ("abun", C.POINTER(C.c_double)),
                ("abun_inflated", C.POINTER(C.c_double))]


def make_meta(meta_ref, abun: dict, abun_inflated: dict | None):
    """NsoMeta + keepalive for a MetaReference and per-species abundance dicts (species order of the reference)."""
    sco = np.ascontiguousarray(meta_ref.species_chrom_off, dtype=np.uint32)
    ab = np.array([abun[sp] for sp in meta_ref.species], dtype=np.float64)
    inf = np.array([abun_inflated[sp] for sp in meta_ref.species], dtype=np.float64) if abun_inflated else np.zeros(len(ab))
    m = NsoMeta()
    m.nspecies = len(meta_ref.species)
    m.species_chrom_off = sco.ctypes.data_as(C.POINTER(C.c_uint32))
    m.abun = ab.ctypes.data_as(C.POINTER(C.c_double))
    m.abun_inflated = inf.ctypes.data_as(C.POINTER(C.c_double))
    return m, (sco, ab, inf)


class NsoLogRow(C.Structure):
    _fields_ = [("pos", C.c_uint32), ("len", C.c_uint32), ("type", C.c_uint32), ("ref_off", C.c_uint32),
                ("new_off", C.c_uint32)]


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(ROOT, "oracle", "libns_oracle.so")
        src = os.path.join(ROOT, "oracle", "ns_oracle.c")
        if not os.path.exists(path) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(path)):
            build()
        L = C.CDLL(path)
        L.nso_log.restype = C.c_double; L.nso_log.argtypes = [C.c_double]
        L.nso_exp.restype = C.c_double; L.nso_exp.argtypes = [C.c_double]
        L.nso_norminv.restype = C.c_double; L.nso_norminv.argtypes = [C.c_double]
        L.nso_pow10m1.restype = C.c_double; L.nso_pow10m1.argtypes = [C.c_double]
        L.nso_ecdf_lookup.restype = C.c_int64
        L.nso_ecdf_lookup.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_uint32, C.c_double, C.c_double]
        L.nso_table_value.restype = C.c_int64
        L.nso_table_value.argtypes = [C.POINTER(C.c_double), C.c_uint32, C.c_double]
        L.nso_trans_pick.restype = C.c_int; L.nso_trans_pick.argtypes = [C.POINTER(C.c_double), C.c_double]
        L.nso_ir_states.restype = C.c_int; L.nso_ir_states.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.nso_run_length.restype = C.c_int64
        L.nso_run_length.argtypes = [C.POINTER(NsModelTables), C.c_int, C.c_double, C.c_double]
        L.nso_kde_sample.restype = C.c_double
        L.nso_kde_sample.argtypes = [C.POINTER(C.c_double), C.c_uint64, C.c_double, C.c_uint32, C.c_uint32, C.c_uint32]
        L.nso_philox.restype = None
        L.nso_philox.argtypes = [C.c_uint32] * 6 + [C.POINTER(C.c_uint32)]
        L.nso_error_list.restype = None
        L.nso_error_list.argtypes = [C.POINTER(NsModelTables), C.c_int64, C.c_int, C.POINTER(NsoDraw), C.c_uint32,
                                     C.c_uint32, C.c_void_p, C.c_uint64, C.POINTER(NsoElist)]
        L.nso_unaligned_error_list.restype = None
        L.nso_unaligned_error_list.argtypes = [C.POINTER(NsModelTables), C.c_int64, C.POINTER(NsoDraw), C.c_uint32,
                                               C.c_uint32, C.c_void_p, C.c_uint64, C.POINTER(NsoElist)]
        L.nso_mutate_read.restype = C.c_int64
        L.nso_mutate_read.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_uint64, C.POINTER(NsoDraw), C.c_uint32,
                                      C.c_uint32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                      C.POINTER(C.c_uint64)]
        L.nso_hp_filter.restype = C.c_uint64
        L.nso_hp_filter.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_uint64, C.c_int64, C.POINTER(C.c_int64)]
        L.nso_mutate_homo.restype = C.c_int64
        L.nso_mutate_homo.argtypes = [C.POINTER(NsModelTables), C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.POINTER(NsoDraw),
                                      C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int64]
        L.nso_hp_mu.restype = C.c_double; L.nso_hp_mu.argtypes = [C.POINTER(NsModelTables), C.c_uint8, C.c_int64]
        L.nso_hp_sigma.restype = C.c_double; L.nso_hp_sigma.argtypes = [C.POINTER(NsModelTables), C.c_uint8, C.c_int64]
        L.nso_assign_species.restype = C.c_uint64
        L.nso_assign_species.argtypes = [C.POINTER(NsoMeta), C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p,
                                         C.POINTER(NsoDraw), C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.nso_extract_meta.restype = C.c_int
        L.nso_extract_meta.argtypes = [C.POINTER(NsoMeta), C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.POINTER(NsoDraw),
                                       C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
        L.nso_generate_meta.restype = C.c_int
        L.nso_generate_meta.argtypes = [C.POINTER(NsModelTables), C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_char_p,
                                        C.POINTER(NsoMeta), C.POINTER(NsParams), C.POINTER(NsoOut), C.c_void_p]
        L.nso_generate_trx.restype = C.c_int
        L.nso_generate_trx.argtypes = [C.POINTER(NsModelTables), C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_char_p,
                                       C.POINTER(NsoTrx), C.POINTER(NsParams), C.POINTER(NsoOut)]
        L.nso_trx_pick.restype = C.c_uint32
        L.nso_trx_pick.argtypes = [C.POINTER(NsoTrx), C.c_double]
        L.nso_generate.restype = C.c_int
        L.nso_generate.argtypes = [C.POINTER(NsModelTables), C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p,
                                   C.c_char_p, C.POINTER(NsParams), C.POINTER(NsoOut)]
        L.nso_qual_value.restype = C.c_uint8
        L.nso_qual_value.argtypes = [C.POINTER(NsModelTables), C.c_int, C.c_uint32]
        L.nso_extract_walk.restype = C.c_int
        L.nso_extract_walk.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_int64, C.POINTER(C.c_uint32),
                                       C.POINTER(C.c_uint64)]
        L.nso_normalise_base.restype = C.c_uint8; L.nso_normalise_base.argtypes = [C.c_uint8]
        L.nso_case_convert.restype = None
        L.nso_case_convert.argtypes = [C.c_void_p, C.c_int64, C.POINTER(NsoDraw), C.c_uint32, C.c_uint32]
        _LIB = L
    return _LIB


def make_tape(u=(), n=(), z=()):
    """Draw source in tape-replay mode; returns (NsoDraw, keepalive)."""
    ua = np.ascontiguousarray(u, dtype=np.float64)
    na = np.ascontiguousarray(n, dtype=np.int64)
    za = np.ascontiguousarray(z, dtype=np.float64)
    d = NsoDraw()
    d.mode = 1
    d.tape_u = ua.ctypes.data_as(C.POINTER(C.c_double)); d.n_u = len(ua)
    d.tape_n = na.ctypes.data_as(C.POINTER(C.c_int64)); d.n_n = len(na)
    d.tape_z = za.ctypes.data_as(C.POINTER(C.c_double)); d.n_z = len(za)
    return d, (ua, na, za)


def make_philox(seed, read):
    d = NsoDraw()
    d.mode = 0
    d.seed = seed
    d.read = read
    return d


def normalise_bases(bases: np.ndarray) -> np.ndarray:
    """Upper-case; non-IUPAC -> N (what ns_set_reference does on the device)."""
    L = lib()
    lut = np.array([L.nso_normalise_base(i) for i in range(256)], dtype=np.uint8)
    return lut[bases]


def generate(model, ref, params: NsParams, *, bytes_per_read=40000, events_per_read=4000):
    """Run the CPU restatement for one batch; returns dict of numpy arrays like the engine's outputs."""
    L = lib()
    t = model.to_c()
    n = int(params.n_reads)
    reads = np.zeros(n, dtype=READ_DTYPE)
    pieces = np.zeros(n * 8 + 64, dtype=PIECE_DTYPE)
    events = np.zeros(n * events_per_read + 1024, dtype=EVENT_DTYPE)
    records = np.zeros(n * bytes_per_read + 4096, dtype=np.uint8)
    errlog = np.zeros((n * bytes_per_read * 3 + 4096) if params.emit_errlog else 16, dtype=np.uint8)
    o = NsoOut()
    o.reads = reads.ctypes.data; o.pieces = pieces.ctypes.data; o.events = events.ctypes.data
    o.cap_pieces = len(pieces); o.cap_events = len(events)
    o.records = records.ctypes.data; o.cap_records = len(records)
    o.errlog = errlog.ctypes.data; o.cap_errlog = len(errlog)
    bases = normalise_bases(ref.bases)
    rc = L.nso_generate(C.byref(t), bases.ctypes.data, ref.chrom_off.ctypes.data, len(ref.names),
                        ref.circular.ctypes.data, ref.names_blob(), C.byref(params), C.byref(o))
    if rc != 0:
        raise RuntimeError("nso_generate failed: %d" % rc)
    return dict(reads=reads, pieces=pieces[:o.n_pieces], events=events[:o.n_events],
                records=records[:o.record_bytes], errlog=errlog[:o.errlog_bytes],
                total_bases=int(o.total_bases), total_ref_bases=int(o.total_ref_bases))


def generate_meta(model, meta_ref, abun: dict, abun_inflated, params: NsParams, *, bytes_per_read=40000, events_per_read=6000):
    """Metagenome batch through the CPU restatement (one call = one reference worker)."""
    L = lib()
    t = model.to_c()
    ref = meta_ref.ref
    n = int(params.n_reads)
    reads = np.zeros(n, dtype=READ_DTYPE)
    pieces = np.zeros(n * 8 + 64, dtype=PIECE_DTYPE)
    events = np.zeros(n * events_per_read + 1024, dtype=EVENT_DTYPE)
    records = np.zeros(n * bytes_per_read + 4096, dtype=np.uint8)
    errlog = np.zeros((n * bytes_per_read * 3 + 4096) if params.emit_errlog else 16, dtype=np.uint8)
    o = NsoOut()
    o.reads = reads.ctypes.data; o.pieces = pieces.ctypes.data; o.events = events.ctypes.data
    o.cap_pieces = len(pieces); o.cap_events = len(events)
    o.records = records.ctypes.data; o.cap_records = len(records)
    o.errlog = errlog.ctypes.data; o.cap_errlog = len(errlog)
    bases = normalise_bases(ref.bases)
    mg, keep = make_meta(meta_ref, abun, abun_inflated)
    sp_bases = np.zeros(len(meta_ref.species), dtype=np.float64)
    rc = L.nso_generate_meta(C.byref(t), bases.ctypes.data, ref.chrom_off.ctypes.data, len(ref.names), ref.circular.ctypes.data,
                             ref.names_blob(), C.byref(mg), C.byref(params), C.byref(o), sp_bases.ctypes.data)
    if rc != 0:
        raise RuntimeError("nso_generate_meta failed: %d" % rc)
    return dict(reads=reads, pieces=pieces[:o.n_pieces], events=events[:o.n_events], records=records[:o.record_bytes],
                errlog=errlog[:o.errlog_bytes], total_bases=int(o.total_bases), total_ref_bases=int(o.total_ref_bases),
                species_bases=sp_bases)


def make_trx(tr, ir=None):
    """tr: nanosim_amd.transcriptome.TranscriptomeReference, ir: nanosim_amd.intron_retention.IntronRetention or None
    -> (NsoTrx, keep-alive list)"""
    ec = np.ascontiguousarray(tr.expr_chrom, dtype=np.uint32)
    cum = np.ascontiguousarray(tr.expr_cum, dtype=np.float64)
    pa = np.ascontiguousarray(tr.polya, dtype=np.uint8)
    x = NsoTrx()
    x.n_expr = len(ec)
    x.expr_chrom = ec.ctypes.data_as(C.POINTER(C.c_uint32)); x.expr_cum = cum.ctypes.data_as(C.POINTER(C.c_double))
    x.polya = pa.ctypes.data_as(C.POINTER(C.c_uint8)) if pa.any() else None
    x.polya_scale = float(tr.polya_scale)
    keep = [ec, cum, pa]
    if ir is not None:
        t = ir.to_c()
        keep += [t, ir]
        x.ir = C.addressof(t)
    return x, keep


def generate_trx(model, tr, params: NsParams, *, ir=None, bytes_per_read=40000, events_per_read=4000):
    """Transcriptome batch through the CPU restatement (aligned / perfect / unaligned by params.kind)."""
    L = lib()
    t = model.to_c()
    ref = tr.ref
    n = int(params.n_reads)
    reads = np.zeros(n, dtype=READ_DTYPE)
    pieces = np.zeros(n * 2 + 64, dtype=PIECE_DTYPE)
    events = np.zeros(n * events_per_read + 1024, dtype=EVENT_DTYPE)
    records = np.zeros(n * bytes_per_read + 4096, dtype=np.uint8)
    errlog = np.zeros((n * bytes_per_read * 3 + 4096) if params.emit_errlog else 16, dtype=np.uint8)
    polya = np.zeros(n + 1, dtype=np.uint16)
    o = NsoOut()
    o.reads = reads.ctypes.data; o.pieces = pieces.ctypes.data; o.events = events.ctypes.data
    o.cap_pieces = len(pieces); o.cap_events = len(events)
    o.records = records.ctypes.data; o.cap_records = len(records)
    o.errlog = errlog.ctypes.data; o.cap_errlog = len(errlog)
    o.polya = polya.ctypes.data
    spliced = np.zeros(n * 4096 + 4096 if ir is not None else 16, dtype=np.uint8)
    o.spliced = spliced.ctypes.data; o.cap_spliced = len(spliced)
    bases = normalise_bases(ref.bases)
    x, keep = make_trx(tr, ir)
    rc = L.nso_generate_trx(C.byref(t), bases.ctypes.data, ref.chrom_off.ctypes.data, len(ref.names), ref.circular.ctypes.data,
                            ref.names_blob(), C.byref(x), C.byref(params), C.byref(o))
    if rc != 0:
        raise RuntimeError("nso_generate_trx failed: %d" % rc)
    return dict(reads=reads, pieces=pieces[:o.n_pieces], events=events[:o.n_events], records=records[:o.record_bytes],
                errlog=errlog[:o.errlog_bytes], total_bases=int(o.total_bases), total_ref_bases=int(o.total_ref_bases), polya=polya[:n],
                spliced=spliced[:o.spliced_bytes])


def _pack_cs(cs_list):
    blobs = [c.encode() if isinstance(c, str) else bytes(c) for c in cs_list]
    off = np.zeros(len(blobs) + 1, dtype=np.uint64)
    np.cumsum([len(b) for b in blobs], out=off[1:])
    return np.frombuffer(b"".join(blobs) + b"\0", dtype=np.uint8), off


def cs_hist(cs_list, cap=2048):
    """The counting loop of src/besthit_to_histogram.py:hist() through the oracle's two-list restatement (nso_cs_hist); same dict as
    nanosim_amd.characterize.count"""
    L = lib()
    L.nso_cs_hist.restype = C.c_int
    L.nso_cs_hist.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32] + [C.c_void_p] * 6
    data, off = _pack_cs(cs_list)
    while True:
        dic = np.zeros((5, 1001), dtype=np.uint64); m2 = np.zeros((cap, cap), dtype=np.uint64)
        err = np.zeros(18, dtype=np.uint64); first = np.zeros(3, dtype=np.uint64); mx = np.zeros(2, dtype=np.uint64)
        rc = L.nso_cs_hist(data.ctypes.data, off.ctypes.data, len(cs_list), cap, dic.ctypes.data, m2.ctypes.data, err.ctypes.data,
                           first.ctypes.data, mx[0:].ctypes.data, mx[1:].ctypes.data)
        if rc:
            raise RuntimeError("nso_cs_hist failed: %d" % rc)
        if not mx[1]:
            break
        cap = 1 << int(mx[0]).bit_length()
    return dict(dic=dic, match_list=m2, error_list=err.reshape(6, 3), first_error=first, max_match=int(mx[0]))


def _pack_maf(pairs):
    """the two lines of every alignment back to back with ONE offset table (both lines of an alignment have the same length)"""
    r = [a.encode() if isinstance(a, str) else bytes(a) for a, _ in pairs]
    q = [b.encode() if isinstance(b, str) else bytes(b) for _, b in pairs]
    assert all(len(a) == len(b) for a, b in zip(r, q))
    off = np.zeros(len(r) + 1, dtype=np.uint64)
    np.cumsum([len(a) for a in r], out=off[1:])
    return np.frombuffer(b"".join(r) + b"\0", dtype=np.uint8), np.frombuffer(b"".join(q) + b"\0", dtype=np.uint8), off


def maf_hist(pairs, cap=2048):
    """hist(prefix, "maf") (src/besthit_to_histogram.py:188-315) through the oracle's column walk (nso_maf_hist); same dict as cs_hist"""
    L = lib()
    L.nso_maf_hist.restype = C.c_int
    L.nso_maf_hist.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32] + [C.c_void_p] * 6
    r, q, off = _pack_maf(pairs)
    while True:
        dic = np.zeros((5, 1001), dtype=np.uint64); m2 = np.zeros((cap, cap), dtype=np.uint64)
        err = np.zeros(18, dtype=np.uint64); first = np.zeros(3, dtype=np.uint64); mx = np.zeros(2, dtype=np.uint64)
        rc = L.nso_maf_hist(r.ctypes.data, q.ctypes.data, off.ctypes.data, len(pairs), cap, dic.ctypes.data, m2.ctypes.data, err.ctypes.data,
                            first.ctypes.data, mx[0:].ctypes.data, mx[1:].ctypes.data)
        if rc:
            raise RuntimeError("nso_maf_hist failed: %d" % rc)
        if not mx[1]:
            break
        cap = 1 << int(mx[0]).bit_length()
    return dict(dic=dic, match_list=m2, error_list=err.reshape(6, 3), first_error=first, max_match=int(mx[0]))


def parse_cs(cs):
    """(list_hist, list_op) of the oracle's parse_cs restatement"""
    L = lib()
    L.nso_parse_cs_lists.restype = C.c_int
    L.nso_parse_cs_lists.argtypes = [C.c_char_p, C.c_uint64, C.c_void_p, C.c_char_p, C.c_uint32, C.c_void_p, C.c_void_p]
    b = cs.encode()
    cap = len(b) + 8
    hist = np.zeros(cap, dtype=np.int64); op = C.create_string_buffer(cap)
    nh, no = C.c_uint32(), C.c_uint32()
    assert L.nso_parse_cs_lists(b, len(b), hist.ctypes.data, op, cap, C.byref(nh), C.byref(no)) == 0
    return [int(x) for x in hist[:nh.value]], [chr(c) for c in op.raw[:no.value]]


def trx_walk_tape(pick_e, pick_y, lengths):
    """nso_trx_walk_tape: the oracle's pick walk over the reference's own picks and look-ups -> (accept, redraw, memo_of, n_samples)"""
    L = lib()
    L.nso_trx_walk_tape.restype = C.c_uint32
    L.nso_trx_walk_tape.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    e = np.ascontiguousarray(pick_e, dtype=np.uint32); y = np.ascontiguousarray(pick_y, dtype=np.int64)
    ln = np.ascontiguousarray(lengths, dtype=np.int64)
    accept = np.zeros(len(e), np.uint8); redraw = np.zeros(len(e), np.uint8); memo = np.zeros(len(e), np.int64)
    n = L.nso_trx_walk_tape(len(e), e.ctypes.data, y.ctypes.data, len(ln), ln.ctypes.data, accept.ctypes.data, redraw.ctypes.data, memo.ctypes.data)
    return accept, redraw, memo, int(n)
