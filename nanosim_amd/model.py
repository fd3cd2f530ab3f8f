"""Host-side model + reference loader: on-disk NanoSim model files -> flat tables.

Re-does the reference's ``read_profile`` (src/simulator.py:244-591) and ``read_ecdf``
(src/simulator.py:194-231) as "parse the same files -> flat numpy arrays" that are handed to the
engine through ``ns_model_tables`` (include/nanosim_amd.h).  No per-read work happens here.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import re
from dataclasses import dataclass, field

import numpy as np

NS_ABI_VERSION = 6
NS_KDE_ALIGNED, NS_KDE_HT, NS_KDE_RATIO, NS_KDE_UNALIGNED, NS_KDE_GAP, NS_KDE_COUNT = 0, 1, 2, 3, 4, 5
NS_Q_NAMES = ("match", "mis", "ins", "ht", "unmapped")
NS_QUAL_LEVELS = 128
NS_HP_MAX_BREAKS = 4
NS_MODEL_HAS_ERRORS, NS_MODEL_HAS_QUALS, NS_MODEL_HAS_HP, NS_MODEL_HAS_CHIMERIC, NS_MODEL_HAS_UNALIGNED, NS_MODEL_HAS_KDE2D = 1, 2, 4, 8, 16, 32
MIX_CAP = 4095
STATE_NAMES = ("start", "mis", "ins", "del", "mis0", "ins0", "del0")


# --------------------------------------------------------------------------------------------------
# ctypes mirror of include/nanosim_amd.h
# --------------------------------------------------------------------------------------------------
class NsKde(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_double)), ("n", C.c_uint64), ("bw", C.c_double)]


class NsHpClass(C.Structure):
    _fields_ = [("konst", C.c_double), ("alpha1", C.c_double), ("n_breaks", C.c_uint32), ("_pad", C.c_uint32),
                ("beta", C.c_double * NS_HP_MAX_BREAKS), ("breakpoint", C.c_double * NS_HP_MAX_BREAKS),
                ("intercept", C.c_double), ("slope", C.c_double)]


class NsModelTables(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32), ("flags", C.c_uint32),
        ("fm_nseg", C.c_uint32), ("_pad0", C.c_uint32),
        ("fm_hi", C.POINTER(C.c_double)), ("fm_vhi", C.POINTER(C.c_double)), ("fm_vlo0", C.c_double),
        ("mm_nbins", C.c_uint32), ("_pad1", C.c_uint32),
        ("mm_bin_lo", C.POINTER(C.c_int64)), ("mm_bin_hi", C.POINTER(C.c_int64)),
        ("mm_seg_off", C.POINTER(C.c_uint32)),
        ("mm_hi", C.POINTER(C.c_double)), ("mm_vhi", C.POINTER(C.c_double)), ("mm_vlo0", C.POINTER(C.c_double)),
        ("trans", (C.c_double * 3) * 7),
        ("mix_w", C.c_double * 3),
        ("mix_n", (C.c_uint32 * 2) * 3),
        ("mix_cdf", (C.POINTER(C.c_double) * 2) * 3),
        ("kde", NsKde * NS_KDE_COUNT),
        ("strandness_rate", C.c_double),
        ("nseg_n", C.c_uint32), ("_pad2", C.c_uint32),
        ("nseg_cdf", C.POINTER(C.c_double)),
        ("qual_thr", (C.c_uint32 * NS_QUAL_LEVELS) * 5),
        ("hp", NsHpClass * 2),
        ("hp_mis_rate", C.c_double),
        ("kde2d_x", C.POINTER(C.c_double)), ("kde2d_y", C.POINTER(C.c_double)), ("kde2d_n", C.c_uint64), ("kde2d_bw", C.c_double),
    ]


class NsParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("first_read", C.c_uint64), ("n_reads", C.c_uint64),
                ("kind", C.c_uint32), ("fastq", C.c_uint32), ("kmer_bias", C.c_uint32), ("chimeric", C.c_uint32),
                ("use_lognormal", C.c_uint32), ("emit_records", C.c_uint32),
                ("min_len", C.c_int64), ("max_len", C.c_int64),
                ("median_len", C.c_double), ("sd_len", C.c_double),
                ("emit_errlog", C.c_uint32), ("meta", C.c_uint32), ("trx", C.c_uint32), ("uracil", C.c_uint32),
                ("model_ir", C.c_uint32), ("reserved0", C.c_uint32)]


class NsIrTables(C.Structure):
    """ns_ir_tables of include/nanosim_amd.h"""
    _fields_ = [("genome", C.c_void_p), ("genome_off", C.c_void_p), ("n_gchrom", C.c_uint32), ("n_items", C.c_uint32),
                ("item_off", C.c_void_p), ("item_type", C.c_void_p), ("item_minus", C.c_void_p), ("item_chrom", C.c_void_p),
                ("item_start", C.c_void_p), ("item_len", C.c_void_p), ("p_no_ir", C.c_double * 3), ("p_ir", C.c_double * 3)]


class NsBatchInfo(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("n_pieces", C.c_uint64), ("n_events", C.c_uint64),
                ("events_used", C.c_uint64), ("record_bytes", C.c_uint64), ("errlog_bytes", C.c_uint64), ("total_bases", C.c_uint64),
                ("total_ref_bases", C.c_uint64), ("n_overflow", C.c_uint64),
                ("ms_total", C.c_double), ("ms_kernel", C.c_double * 8), ("spliced_bytes", C.c_uint64),
                ("n_range_redraws", C.c_uint64)]


EVENT_DTYPE = np.dtype([("pos", "<u4"), ("info", "<u4")])
PIECE_DTYPE = np.dtype([("ref_gpos", "<u8"), ("ev_off", "<u8"), ("chrom", "<u4"), ("pos", "<u4"),
                        ("ref_len", "<u4"), ("out_len", "<u4"), ("n_ev", "<u4"), ("kind", "<u4")])
READ_DTYPE = np.dtype([("rec_off", "<u8"), ("piece_off", "<u4"), ("n_pieces", "<u2"), ("reversed", "u1"),
                       ("flags", "u1"), ("head", "<u4"), ("tail", "<u4"), ("seq_len", "<u4"), ("attempts", "<u4")])
assert EVENT_DTYPE.itemsize == 8 and PIECE_DTYPE.itemsize == 40 and READ_DTYPE.itemsize == 32
NS_EV_SHIFT_BIAS = 131072


def ev_len(info):
    return np.asarray(info, dtype=np.uint32) & 0xFFF


def ev_type(info):
    return (np.asarray(info, dtype=np.uint32) >> 12) & 3


def ev_shift(info):
    return (np.asarray(info, dtype=np.uint32) >> 14).astype(np.int64) - NS_EV_SHIFT_BIAS


def ev_pack(length, ty, shift):
    return (int(length) & 0xFFF) | (int(ty) & 3) << 12 | (int(shift) + NS_EV_SHIFT_BIAS) << 14


# --------------------------------------------------------------------------------------------------
# read_ecdf (src/simulator.py:194-231)
# --------------------------------------------------------------------------------------------------
@dataclass
class EcdfColumn:
    lo: int            # bin boundaries from the header (k1[0], k1[1])
    hi_bin: int
    hi: np.ndarray     # upper CDF edge of each segment (lower edge = previous hi, first = 0.0)
    vhi: np.ndarray    # upper value of each segment (lower value = previous vhi, first = vlo0)
    vlo0: float

    def segments(self):
        """[(cdf_lo, cdf_hi, v_lo, v_hi)] exactly as the reference's nested dict would list them."""
        out, plo, vlo = [], 0.0, self.vlo0
        for h, v in zip(self.hi, self.vhi):
            out.append((plo, float(h), vlo, float(v)))
            plo, vlo = float(h), float(v)
        return out


def read_ecdf(path: str) -> list[EcdfColumn]:
    """Columns in the order error_list iterates them (dict insertion order = header order,
    src/simulator.py:1891-1893); column i's numbers are attached to the i-th *sorted* key exactly as
    the reference does (src/simulator.py:206,217-222)."""
    with open(path) as f:
        header_info = f.readline().strip().split()
        keys = []
        for item in header_info[1:]:
            b = item.split("-")
            keys.append((int(b[0]), int(b[1])))
        lanes = len(keys)
        ecdf_key = sorted(keys)
        l_prob = [0.0] * lanes
        l_ratio = [0.0] * lanes
        segs = [[] for _ in range(lanes)]   # (p_lo, p_hi, v_lo, v_hi)
        ratio = None
        for line in f:
            new = line.strip().split("\t")
            if len(new) < 2:
                continue
            ratio = [float(x) for x in new[0].split("-")]
            prob = [float(x) for x in new[1:]]
            for i in range(lanes):
                if prob[i] == l_prob[i]:
                    continue
                if l_prob[i] != 0:
                    segs[i].append([l_prob[i], prob[i], l_ratio[i], ratio[1]])
                else:
                    segs[i].append([l_prob[i], prob[i],
                                    max(l_ratio[i], ratio[1] - 10 * (ratio[1] - ratio[0])), ratio[1]])
                l_ratio[i] = ratio[1]
                l_prob[i] = prob[i]
        for i in range(lanes):
            if segs[i]:
                segs[i][-1][3] = ratio[1]      # S:226-229: last segment reaches the last row of the file
    by_key = {ecdf_key[i]: segs[i] for i in range(lanes)}
    cols = []
    for k in dict.fromkeys(keys):              # header (insertion) order, duplicates collapse like a dict
        s = by_key[k]
        if not s:
            raise ValueError("%s: ECDF column %d-%d has no probability mass" % (path, k[0], k[1]))
        cols.append(EcdfColumn(k[0], k[1], np.array([x[1] for x in s], dtype=np.float64),
                               np.array([x[3] for x in s], dtype=np.float64), float(s[0][2])))
    return cols


# --------------------------------------------------------------------------------------------------
# inverse-CDF tables of the run-length mixtures (src/mixed_model.py:41-63)
# --------------------------------------------------------------------------------------------------
def _trim(cdf: np.ndarray) -> np.ndarray:
    """Keep entries until the remaining tail mass is below 2^-60 (or the cap)."""
    tail = 1.0 - cdf
    idx = np.nonzero(tail < 2.0 ** -60)[0]
    n = int(idx[0]) + 1 if idx.size else len(cdf)
    return np.ascontiguousarray(cdf[:max(n, 1)], dtype=np.float64)


def poisson_plus1_cdf(lam: float) -> np.ndarray:
    """cdf[j] = P(Poisson(lam)+1 <= j+1) = P(Poisson <= j)   (src/mixed_model.py:46)."""
    j = np.arange(MIX_CAP, dtype=np.float64)
    if lam <= 0:
        return np.ones(1)
    logpmf = -lam + j * math.log(lam) - np.array([math.lgamma(x + 1.0) for x in j])
    return _trim(np.minimum(np.cumsum(np.exp(logpmf)), 1.0))


def geometric_cdf(p: float, shift: int) -> np.ndarray:
    """Geometric(p) on {1,2,..}; value = G - shift with 0 mapped to 1 (src/mixed_model.py:48,58-61).
    cdf[j] = P(value <= j+1) = P(G <= j+1+shift) = 1-(1-p)^(j+1+shift)."""
    if p >= 1.0:
        return np.ones(1)
    j = np.arange(MIX_CAP, dtype=np.float64)
    return _trim(-np.expm1((j + 1.0 + shift) * math.log1p(-p)))


def weibull_ceil_cdf(lam: float, k: float) -> np.ndarray:
    """value = int(round(ceil(lam*Weibull(k)))) with 0 -> 1 (src/mixed_model.py:56,60-61).
    cdf[j] = P(value <= j+1) = 1 - exp(-((j+1)/lam)^k)."""
    v = np.arange(1, MIX_CAP + 1, dtype=np.float64)
    return _trim(-np.expm1(-np.power(v / lam, k)))


def lognorm_cdf(x: float, s: float, mu: float) -> float:
    if x <= 0:
        return 0.0
    return 0.5 * (1.0 + math.erf((math.log(x) - mu) / (s * math.sqrt(2.0))))


def quality_thresholds(sd: float, loc: float, mu: float) -> np.ndarray:
    """Integer quality sampler of predict_base_qualities (src/model_base_qualities.py:9-20,120-130):
    q = int64(trunc_lognorm[1,93).ppf(U) + loc).  thr[j] = round(65536*P(q <= j)), j = 0..127."""
    fa, fb = lognorm_cdf(1.0, sd, mu), lognorm_cdf(93.0, sd, mu)
    thr = np.zeros(NS_QUAL_LEVELS, dtype=np.uint32)
    for j in range(NS_QUAL_LEVELS):
        x = min(max(j + 1.0 - loc, 1.0), 93.0)     # q <= j  <=>  x + loc < j+1
        c = (lognorm_cdf(x, sd, mu) - fa) / (fb - fa)
        thr[j] = int(math.floor(65536.0 * c + 0.5))
    return thr


def snap_quality_thresholds(thr: np.ndarray) -> np.ndarray:
    """The engine looks a 16-bit draw h up in 1024 buckets of 64 values (count of thresholds at or below the bucket start, plus
    ONE compare against the only threshold strictly inside the bucket).  Where several thresholds share a bucket — quality levels
    of mass < 2^-10 each, in the two tails — all but one are moved to the nearer bucket boundary (CDF displacement <= 32/65536 at
    those levels only); which one stays is chosen to minimise the total displacement.  Every consumer (engine, oracle) gets the
    snapped table, so q = #{j : h >= thr[j]} holds exactly for all of them."""
    thr = np.asarray(thr, dtype=np.int64).copy()
    b = 0
    while b < 1024:
        lo, hi = 64 * b, 64 * b + 63
        inside = np.nonzero((thr > lo) & (thr <= hi))[0]
        if len(inside) > 1:
            t = thr[inside]
            best, best_cost = 0, None
            for keep in range(len(inside)):
                cost = int(np.sum(t[:keep] - lo) + np.sum(lo + 64 - t[keep + 1:]))
                # thresholds below the kept one go down, those above it go up (monotone)
                if best_cost is None or cost < best_cost:
                    best, best_cost = keep, cost
            thr[inside[:best]] = lo
            thr[inside[best + 1:]] = lo + 64
        b += 1
    return thr.astype(np.uint32)


def quality_pmf(sd: float, loc: float, mu: float) -> np.ndarray:
    thr = quality_thresholds(sd, loc, mu).astype(np.float64) / 65536.0
    return np.diff(np.concatenate([[0.0], thr]))


# --------------------------------------------------------------------------------------------------
@dataclass
class Model:
    prefix: str
    perfect: bool = False
    error_par: dict = field(default_factory=dict)          # S:473-484
    trans: np.ndarray | None = None                        # 7x3 (a, a+b, 1-c), S:486-495
    trans_raw: np.ndarray | None = None
    first_match: EcdfColumn | None = None
    match_markov: list[EcdfColumn] = field(default_factory=list)
    mix_cdf: list = field(default_factory=list)            # [type][component]
    mix_w: np.ndarray | None = None
    kde: dict = field(default_factory=dict)                # index -> (data, bw)
    alignment_rate: float | None = None                    # None == "100%"
    strandness_rate: float = 0.5
    segment_mean: float | None = None
    abun_inflation: float | None = None
    nseg_cdf: np.ndarray | None = None
    quals: dict = field(default_factory=dict)              # type -> (sd, loc, mu)
    qual_thr: np.ndarray | None = None
    hp: dict = field(default_factory=dict)
    hp_mis_rate: float = 0.0
    kde2d: tuple | None = None                             # transcriptome: (x sorted, y, bandwidth) of _aligned_region_2d
    _keep: list = field(default_factory=list, repr=False)

    # -- counts (S:535-542) ---------------------------------------------------
    def split_counts(self, number: int) -> tuple[int, int]:
        if self.perfect or self.alignment_rate is None:
            return number, 0
        r = self.alignment_rate
        n_al = int(round(number * r / (r + 1)))
        return n_al, number - n_al

    # -- C view ------------------------------------------------------------------------------------
    def to_c(self) -> NsModelTables:
        t = NsModelTables()
        keep = self._keep = []

        def dptr(a, ctype=C.c_double, dtype=np.float64):
            a = np.ascontiguousarray(a, dtype=dtype)
            keep.append(a)
            return a.ctypes.data_as(C.POINTER(ctype))

        t.abi_version = NS_ABI_VERSION
        flags = 0
        if not self.perfect:
            flags |= NS_MODEL_HAS_ERRORS
            fm = self.first_match
            t.fm_nseg = len(fm.hi)
            t.fm_hi, t.fm_vhi, t.fm_vlo0 = dptr(fm.hi), dptr(fm.vhi), fm.vlo0
            mm = self.match_markov
            t.mm_nbins = len(mm)
            t.mm_bin_lo = dptr([c.lo for c in mm], C.c_int64, np.int64)
            t.mm_bin_hi = dptr([c.hi_bin for c in mm], C.c_int64, np.int64)
            off = np.concatenate([[0], np.cumsum([len(c.hi) for c in mm])]).astype(np.uint32)
            t.mm_seg_off = dptr(off, C.c_uint32, np.uint32)
            t.mm_hi = dptr(np.concatenate([c.hi for c in mm]))
            t.mm_vhi = dptr(np.concatenate([c.vhi for c in mm]))
            t.mm_vlo0 = dptr([c.vlo0 for c in mm])
            for s in range(7):
                for j in range(3):
                    t.trans[s][j] = float(self.trans[s, j])
            for ty in range(3):
                t.mix_w[ty] = float(self.mix_w[ty])
                for comp in range(2):
                    t.mix_n[ty][comp] = len(self.mix_cdf[ty][comp])
                    t.mix_cdf[ty][comp] = dptr(self.mix_cdf[ty][comp])
        for idx, (data, bw) in self.kde.items():
            t.kde[idx].data = dptr(data)
            t.kde[idx].n = len(data)
            t.kde[idx].bw = float(bw)
        if NS_KDE_UNALIGNED in self.kde:
            flags |= NS_MODEL_HAS_UNALIGNED
        t.strandness_rate = float(self.strandness_rate)
        if self.nseg_cdf is not None and NS_KDE_GAP in self.kde:
            flags |= NS_MODEL_HAS_CHIMERIC
            t.nseg_n = len(self.nseg_cdf)
            t.nseg_cdf = dptr(self.nseg_cdf)
        if self.qual_thr is not None:
            flags |= NS_MODEL_HAS_QUALS
            for c in range(5):
                for j in range(NS_QUAL_LEVELS):
                    t.qual_thr[c][j] = int(self.qual_thr[c, j])
        if self.hp:
            flags |= NS_MODEL_HAS_HP
            for i, base in enumerate(("AT", "CG")):
                h = self.hp[base]
                t.hp[i].konst, t.hp[i].alpha1 = h["const"], h["alpha1"]
                t.hp[i].n_breaks = len(h["breakpoints"])
                for j, (b, bp) in enumerate(zip(h["betas"], h["breakpoints"])):
                    t.hp[i].beta[j], t.hp[i].breakpoint[j] = b, bp
                t.hp[i].intercept, t.hp[i].slope = h["intercept"], h["slope"]
            t.hp_mis_rate = self.hp_mis_rate
        if self.kde2d is not None:
            flags |= NS_MODEL_HAS_KDE2D
            t.kde2d_x, t.kde2d_y = dptr(self.kde2d[0]), dptr(self.kde2d[1])
            t.kde2d_n, t.kde2d_bw = len(self.kde2d[0]), float(self.kde2d[2])
        t.flags = flags
        return t


def _kde_pickle_tolerant(path: str):
    """(training data, bandwidth) of a pickled sklearn KernelDensity WITHOUT building the sklearn objects: the pre-trained NanoSim
    models were pickled with scikit-learn 0.22 (README.md:41) and do not load under a current scikit-learn (moved modules, changed
    Cython tree layout).  Every sklearn symbol of the pickle becomes a stub that just keeps its constructor arguments / state; the
    KDTree's state tuple starts with the training matrix, the estimator's state dict holds `bandwidth` (`bandwidth_` from 1.2 on)."""
    from joblib.numpy_pickle import NumpyUnpickler

    class _Stub:
        def __init__(self, *a, **k):
            self.args, self.state = a, None

        def __setstate__(self, st):
            self.state = st

    class _Unpickler(NumpyUnpickler):
        def find_class(self, module, name):
            if module.split(".")[0] == "sklearn":
                return type(name, (_Stub,), {})
            return super().find_class(module, name)

    import inspect
    with open(path, "rb") as f:
        if "ensure_native_byte_order" in inspect.signature(NumpyUnpickler.__init__).parameters:      # joblib >= 1.3
            est = _Unpickler(path, f, True, mmap_mode=None).load()
        else:
            est = _Unpickler(path, f, mmap_mode=None).load()
    st = est.state if isinstance(est.state, dict) else est.__dict__
    tree = st["tree_"]
    data = tree.state[0] if isinstance(tree.state, (tuple, list)) else tree.state["data"]
    bw = st.get("bandwidth_", None)
    if bw is None:
        bw = st["bandwidth"]
    return np.asarray(data, dtype=np.float64), float(bw)


def _kde_pickle(path: str):
    """(training matrix, bandwidth) of `<prefix>_<name>.pkl` (S:545-567): through sklearn when the pickle loads, else tolerant"""
    if os.environ.get("NS_KDE_TOLERANT") != "1":
        try:
            import joblib                      # only needed for genuine trained models
            kde = joblib.load(path)
            data = np.asarray(kde.tree_.data, dtype=np.float64)
            bw = getattr(kde, "bandwidth_", None)
            if bw is None:
                bw = kde.bandwidth
            return data, float(bw)
        except Exception:                      # wrong scikit-learn for this pickle
            pass
    return _kde_pickle_tolerant(path)


def _load_kde(prefix: str, name: str, npz):
    if npz is not None and name + "_data" in npz:
        return np.asarray(npz[name + "_data"], dtype=np.float64), float(npz[name + "_bw"])
    path = prefix + "_" + name + ".pkl"
    if not os.path.exists(path):
        return None
    data, bw = _kde_pickle(path)
    if data.ndim == 2 and data.shape[1] != 1:
        raise ValueError(path + ": only 1-D KDEs are supported on this path")
    return np.ascontiguousarray(data.reshape(-1)), bw


def _load_kde2d(prefix: str, npz):
    """(x sorted ascending, y in the same order, bandwidth) of the 2-D KDE `_aligned_region_2d` (S:561-565)"""
    if npz is not None and "aligned_region_2d_data" in npz:
        data, bw = np.asarray(npz["aligned_region_2d_data"], dtype=np.float64), float(npz["aligned_region_2d_bw"])
    else:
        path = prefix + "_aligned_region_2d.pkl"
        if not os.path.exists(path):
            return None
        data, bw = _kde_pickle(path)
    order = np.argsort(data[:, 0], kind="stable")
    return np.ascontiguousarray(data[order, 0]), np.ascontiguousarray(data[order, 1]), bw


def load_model(prefix: str, *, perfect: bool = False, strandness: float | None = None, chimeric: bool = False,
               homopolymer: bool = False, fastq: bool = False, need_unaligned: bool = True, transcriptome: bool = False) -> Model:
    """Mirror of read_profile()'s model half for genome/metagenome mode (src/simulator.py:268-275,465-591)."""
    m = Model(prefix=prefix, perfect=perfect)
    if strandness is None:
        with open(prefix + "_strandness_rate") as f:                       # S:270-273
            m.strandness_rate = float(f.readline().split("\t")[1])
    else:
        m.strandness_rate = float(strandness)
    npz = np.load(prefix + "_kde.npz") if os.path.exists(prefix + "_kde.npz") else None

    if not perfect:
        with open(prefix + "_model_profile") as f:                         # S:473-484
            f.readline()
            for line in f:
                new_line = line.strip().split("\t")
                if len(new_line) < 5:
                    continue
                if "mismatch" in line:
                    m.error_par["mis"] = [float(x) for x in new_line[1:]]
                elif "insertion" in line:
                    m.error_par["ins"] = [float(x) for x in new_line[1:]]
                else:
                    m.error_par["del"] = [float(x) for x in new_line[1:]]
        raw = {}
        with open(prefix + "_error_markov_model") as f:                    # S:486-495
            f.readline()
            for line in f:
                info = line.strip().split()
                if len(info) >= 4:
                    raw[info[0]] = (float(info[1]), float(info[2]), float(info[3]))
        m.trans_raw = np.array([raw[s] for s in STATE_NAMES], dtype=np.float64)
        m.trans = np.array([[a, a + b, 1 - c] for a, b, c in m.trans_raw], dtype=np.float64)
        m.first_match = read_ecdf(prefix + "_first_match.hist")[0]         # S:497-498 (first key only, S:1844)
        m.match_markov = read_ecdf(prefix + "_match_markov_model")         # S:500-501
        lam, _, p, w = m.error_par["mis"]
        m.mix_cdf = [[poisson_plus1_cdf(lam), geometric_cdf(p, 0)]]
        ws = [w]
        for ty in ("ins", "del"):
            lam, k, p, w = m.error_par[ty]
            m.mix_cdf.append([weibull_ceil_cdf(lam, k), geometric_cdf(p, 1)])
            ws.append(w)
        m.mix_w = np.array(ws, dtype=np.float64)
        if homopolymer:                                                     # S:504-529
            with open(prefix + "_hp_lengths_model_parameters.tsv") as f:
                m.hp_mis_rate = float(re.search(r"\d+\.?\d*", next(f))[0])
                col_names = next(f).strip().split("\t")
                for line in f:
                    fields = line.strip().split("\t")
                    if len(fields) < 2:
                        continue
                    h = {"betas": [], "breakpoints": []}
                    for i, cn in enumerate(col_names):
                        if i == 0:
                            continue
                        v = float(fields[i])
                        if "breakpoint" in cn:
                            h["breakpoints"].append(v)
                        elif "beta" in cn:
                            h["betas"].append(v)
                        else:
                            h[cn] = v
                    if len(h["betas"]) != len(h["breakpoints"]) or len(h["betas"]) > NS_HP_MAX_BREAKS:
                        raise ValueError("unsupported piecewise parameter set in hp model")
                    m.hp[fields[0]] = h
        with open(prefix + "_reads_alignment_rate") as f:                  # S:535-542
            rate = f.readline().strip().split("\t")[1]
            m.alignment_rate = None if rate == "100%" else float(rate)
        if need_unaligned and m.alignment_rate is not None:
            k = _load_kde(prefix, "unaligned_length", npz)                  # S:544-545
            if k is not None:
                m.kde[NS_KDE_UNALIGNED] = k
    m.kde[NS_KDE_HT] = _load_kde(prefix, "ht_length", npz)                  # S:552
    m.kde[NS_KDE_RATIO] = _load_kde(prefix, "ht_ratio", npz)                # S:555
    m.kde[NS_KDE_ALIGNED] = _load_kde(prefix, "aligned_reads" if perfect else "aligned_region", npz)  # S:559-567
    if transcriptome:                                                       # S:559-565: the 2-D KDE replaces the aligned-length KDE
        m.kde2d = _load_kde2d(prefix, npz)
        if m.kde2d is None:
            raise FileNotFoundError("missing 2-D KDE (_aligned_region_2d) for model prefix " + prefix)
        if m.kde[NS_KDE_ALIGNED] is None:
            del m.kde[NS_KDE_ALIGNED]
    for k in (NS_KDE_HT, NS_KDE_RATIO) + (() if transcriptome else (NS_KDE_ALIGNED,)):
        if m.kde[k] is None:
            raise FileNotFoundError("missing KDE for model prefix " + prefix)
    if chimeric:                                                            # S:571-577
        with open(prefix + "_chimeric_info") as f:
            m.segment_mean = float(f.readline().split("\t")[1])
            second = f.readline()
            if second.strip():
                m.abun_inflation = float(second.split("\t")[1])
        m.kde[NS_KDE_GAP] = _load_kde(prefix, "gap_length", npz)
        m.nseg_cdf = geometric_cdf(1.0 / m.segment_mean, 0)                 # np.random.geometric(1/segment_mean), S:1277
    if fastq:                                                               # S:580-591
        with open(prefix + "_base_qualities_model_parameters.tsv") as f:
            next(f)
            for line in f:
                fields = line.split("\t")
                if len(fields) >= 4:
                    m.quals[fields[0]] = (float(fields[1]), float(fields[2]), float(fields[3]))
        m.qual_thr = np.stack([snap_quality_thresholds(quality_thresholds(*m.quals[nm])) for nm in NS_Q_NAMES])
    return m


# --------------------------------------------------------------------------------------------------
# reference genome (readfq + name normalisation, src/simulator.py:341-349, 709-740)
# --------------------------------------------------------------------------------------------------
@dataclass
class Reference:
    names: list[str]
    bases: np.ndarray            # uint8, concatenation of all chromosomes as in the FASTA (any case)
    chrom_off: np.ndarray        # uint64 [nchrom+1]
    circular: np.ndarray         # uint8 [nchrom]

    @property
    def genome_len(self) -> int:
        return int(self.chrom_off[-1])

    @property
    def max_chrom(self) -> int:
        return int(np.diff(self.chrom_off.astype(np.int64)).max())

    def names_blob(self) -> bytes:
        return b"".join(n.encode() + b"\0" for n in self.names)

    def chrom(self, i: int) -> np.ndarray:
        return self.bases[int(self.chrom_off[i]):int(self.chrom_off[i + 1])]


def normalise_name(header_name: str) -> str:
    info = re.split(r"[_\s]\s*", header_name)          # S:344
    return "-".join(info).split(".")[0]                 # S:345-347


def read_fasta(path: str, dna_type: str = "linear", raw_names: bool = False) -> Reference:
    """FASTA/FASTQ reader with readfq's record semantics (src/simulator.py:709-740) for FASTA input and
    the chromosome-name normalisation of S:344-347 (raw_names: the header up to the first white space instead)."""
    data = np.fromfile(path, dtype=np.uint8)
    if data.size == 0:
        raise ValueError("empty reference file " + path)
    raw = data.tobytes()
    if raw[:1] == b"@":
        return _read_fastq_slow(path, dna_type)
    names, chunks = [], []
    pos = 0
    n = len(raw)
    # header lines start at file start or right after a newline
    starts = [0] if raw[:1] == b">" else []
    nl_gt = np.nonzero((data[:-1] == 10) & (data[1:] == ord(">")))[0] + 1
    starts.extend(int(x) for x in nl_gt)
    for si, s in enumerate(starts):
        e = raw.find(b"\n", s)
        if e < 0:
            e = n
        hdr = raw[s + 1:e].decode()
        name = hdr.partition(" ")[0]                                        # S:719
        body_end = starts[si + 1] if si + 1 < len(starts) else n
        body = data[e + 1:body_end]
        body = body[(body != 10) & (body != 13)]
        names.append(hdr.split()[0] if raw_names and hdr.split() else normalise_name(name))
        chunks.append(body)
    if not names:
        raise ValueError("no FASTA records in " + path)
    return make_reference(names, chunks, dna_type)


def _read_fastq_slow(path: str, dna_type: str) -> Reference:
    names, chunks = [], []
    with open(path) as f:
        lines = f.read().split("\n")
    i = 0
    while i < len(lines):
        if lines[i][:1] == "@":
            names.append(normalise_name(lines[i][1:].partition(" ")[0]))
            chunks.append(np.frombuffer(lines[i + 1].encode(), dtype=np.uint8))
            i += 4
        else:
            i += 1
    return make_reference(names, chunks, dna_type)


def make_reference(names: list[str], chunks: list[np.ndarray], dna_type: str = "linear",
                   circular: list[bool] | None = None) -> Reference:
    # duplicate normalised names overwrite the earlier sequence but keep its dict position (S:346)
    order, seqs = [], {}
    for nm, ch in zip(names, chunks):
        if nm not in seqs:
            order.append(nm)
        seqs[nm] = ch
    lens = np.array([len(seqs[nm]) for nm in order], dtype=np.uint64)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    bases = np.concatenate([seqs[nm] for nm in order]).astype(np.uint8) if order else np.zeros(0, np.uint8)
    if circular is None:
        circ = np.full(len(order), 1 if dna_type == "circular" else 0, dtype=np.uint8)
    else:
        circ = np.array([1 if c else 0 for c in circular], dtype=np.uint8)
    return Reference(order, np.ascontiguousarray(bases), off, circ)
