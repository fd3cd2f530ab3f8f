// ns_device.h — device-side restatement of the per-read generation functions (gfx950).
// error_list / unaligned_error_list / letters / qualities as __device__ functions shared by the kernels
// in nanosim_amd.hip.  Reference lines: S: = src/simulator.py of bcgsc/NanoSim v3.2.2.
#pragma once
#include "ns_rng.h"
#include "../../include/nanosim_amd.h"

struct ChainTab {                 // offsets are in 8-byte words from the start of the blob (ns_pack.h describes the image)
    uint32_t n_words;
    uint32_t trans;               // 21 thresholds: rows start,mis,ins,del,mis0,ins0,del0 x (a, a+b, -)
    uint32_t mix_w;               // 3 thresholds
    uint32_t mix_rec;             // the run-length tables' (offset, length, guide) by 2 * type + component as records of two words
                                  // {offset | length << 32, guide}: indexed by a per-thread type, three fields of this struct would be
                                  // three vector loads from the kernel-argument segment per event
    uint32_t fm_hi, fm_vhi, fm_n, fm_guide;
    uint32_t mm_gv;               // ONE word per ECDF segment: threshold | class | value (ecdf_lookup_gv) — in the LDS part the HOT PREFIX of every
                                  // match-length column (round 6): the segments a draw reaches with probability >= 1 - 2^-tail_bits; a draw beyond
                                  // them takes the full column in global memory (mm_gv_full): a trained model's 15 x 1 500 segments are 180 KB
    uint32_t pm_lut;              // the (prefix) column of a previous match < 256 in one word {first segment | segments << 32 | bin << 56}
    uint32_t sub2;                // the steps of the interpolation inside the narrow segments (of the prefix columns)
    uint32_t pm_bin;              // the prefix column of every bin, same word format (previous matches >= 256 find their bin by the walk of S:1891-1893)
    uint32_t n_words_lds;         // the blob up to here goes to LDS (k_chain<LDS>); what lies behind it stays in global memory:
    uint32_t fm_gv, mm_gv_full, pm_full, sub2_full;   // the first-match column (one look-up per piece), the FULL match-length columns, their
                                  // {first segment | segments | bin} words and step lists; then the fp64 tables
    uint32_t mm_g16;              // per match-length column a 65 536-cell guide over its FULL one-word segments (uint16: segments passed by the
                                  // smallest draw of the cell) — the wave-per-read chain evaluates every column for every future iteration
                                  // (coop_error_list): guide -> two segment words, two dependent reads per column instead of a walk
    uint32_t int_image;           // every value edge is a whole number: the integer image is valid (chain_error_list), in LDS or not
    uint32_t tail_bits;           // the prefix columns end where 2^-tail_bits of the probability is left (0: they are the full columns)
    uint32_t n_words_mix;         // ... and up to here (trans, mix_w, the run-length tables, mix_rec) is all unaligned_error_list reads:
                                  // the LDS image of the wave-per-read unaligned chain (k_chain<true, true>)
    uint32_t mm_nbins, mm_bin, mm_bin_lut, mm_seg_off, mm_hi, mm_vhi, mm_vlo0, mm_guide;
    double fm_vlo0;
};

struct DevModel {
    uint32_t flags;
    ChainTab ct;                  // packed chain tables (ns_chain.h)
    const uint64_t *chain_blob;
    uint32_t fm_nseg;
    const double *fm_hi, *fm_vhi;
    double fm_vlo0;
    uint32_t mm_nbins;
    const int64_t *mm_bin_lo, *mm_bin_hi;
    const uint32_t *mm_seg_off;
    const double *mm_hi, *mm_vhi, *mm_vlo0;
    double trans[7][3];
    double mix_w[3];
    uint32_t mix_n[3][2];
    const double *mix_cdf[3][2];
    ns_kde kde[NS_KDE_COUNT];
    double strandness_rate;
    uint32_t nseg_n;
    const double *nseg_cdf;
    const uint32_t *qual_thr;     // [NS_Q_COUNT][NS_QUAL_LEVELS]
    const uint16_t *qual_lut;     // [NS_Q_COUNT][1024], bucket b = h >> 6: bits 7-13 #{j : thr[j] <= 64 b}; bits 0-6 = 128 - (offset inside
                                  // the bucket of its only threshold), 64 if it has none; bit 15: several thresholds inside (walk them)
    ns_hp_class hp[2];
    double hp_mis_rate;
    const double *kde2d_x, *kde2d_y;      // transcriptome: 2-D KDE training points sorted by transcript length
    uint64_t kde2d_n;
    double kde2d_bw;
};

// expression view of the reference transcripts (ns_set_transcriptome)
struct DevTrx {
    uint32_t n_expr;
    const uint32_t *expr_chrom;
    const double *expr_cum;
    const uint8_t *polya;                 // [nchrom] or nullptr
    double polya_scale;
};

struct DevRef {
    const uint8_t *bases;         // normalised: upper-case ASCII, IUPAC codes kept, everything else N
    const uint64_t *chrom_off;    // [nchrom+1]
    const uint8_t *circular;      // [nchrom]
    const char *names;            // NUL-separated
    const uint32_t *name_off;     // [nchrom+1] offsets into names
    uint32_t nchrom;
    const uint8_t *spliced;       // intron retention: the batch's splice arena (pieces with ref_gpos >= NS_SPLICED_BASE read from it)
};

// ---- table look-ups ------------------------------------------------------------------------------------
// ECDF look-up of S:1845-1849 / S:1895-1898: segment with lo < p <= hi, linear interpolation, floor.
__device__ __forceinline__ int64_t ecdf_lookup(const double *__restrict__ hi, const double *__restrict__ vhi,
                                               uint32_t n, double vlo0, double p) {
    uint32_t lo_i = 0, hi_i = n;
    while (lo_i < hi_i) {
        uint32_t mid = (lo_i + hi_i) >> 1;
        if (p <= hi[mid]) hi_i = mid; else lo_i = mid + 1;
    }
    uint32_t s = lo_i;
    if (s >= n) { s = n - 1; p = hi[s]; }
    double plo = s ? hi[s - 1] : 0.0;
    double vlo = s ? vhi[s - 1] : vlo0;
    return (int64_t)floor((p - plo) / (hi[s] - plo) * (vhi[s] - vlo) + vlo);
}

__device__ __forceinline__ int64_t table_value(const double *__restrict__ cdf, uint32_t n, double p) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if (p <= cdf[mid]) hi = mid; else lo = mid + 1;
    }
    if (lo >= n) lo = n - 1;
    return (int64_t)lo + 1;
}

// Integer form of a probability compare.  A 32-bit draw u stands for p = (u + 0.5) 2^-32, so for a table value t (fp64)
//   p <  t   <=>   u <  ceil(t 2^32 - 0.5)        =: ns_thr_lt(t)
//   p >  t   <=>   u >= floor(t 2^32 - 0.5) + 1   =: ns_thr_gt(t)
// exactly (t 2^32 and the - 0.5 are exact in fp64), with thresholds in [0, 2^32] — hence 64-bit.  The chains compare the draw with
// these instead of converting it to fp64 first (three fp64 operations per draw and an fp64 compare per table step); the host
// (ns_load_model) builds them, the oracle keeps the fp64 compares, and the bit-exact parity tests hold the two together.
__host__ __device__ inline uint64_t ns_thr_lt(double t) {
    if (!(t > 0.0)) return 0;
    if (t >= 1.0) return 1ull << 32;
    const double y = ceil(t * 4294967296.0 - 0.5);
    return y <= 0.0 ? 0ull : (uint64_t)y;
}
__host__ __device__ inline uint64_t ns_thr_gt(double t) {
    if (t < 0.0) return 0;
    if (t >= 1.0) return 1ull << 32;
    const double y = floor(t * 4294967296.0 - 0.5) + 1.0;
    return y <= 0.0 ? 0ull : (uint64_t)y;
}

// S:1860-1864 on the integer thresholds of a transition row (T0 = ns_thr_lt(a), T1 = ns_thr_lt(a + b)): the intervals are tested in
// the order mis [0, a), ins [a, a + b), del [1 - c, 1); a draw that falls between a + b and 1 - c (rounding gap) takes del
// (DESIGN.md section 5.5), so everything that is neither mis nor ins is del
NS_DEV int trans_pick_u(const uint64_t *row, uint32_t u) {
    return (uint64_t)u < row[0] ? NS_MIS : (uint64_t)u < row[1] ? NS_INS : NS_DEL;
}

// ---- lengths ---------------------------------------------------------------------------------------------
// KernelDensity.sample (call site S:235): i = floor(U*n); x = N(data[i], bw)
__device__ __forceinline__ double kde_sample(const ns_kde &k, const u32x4 &w) {
    uint64_t i = (uint64_t)(u53_to_p(w.x, w.y) * (double)k.n);
    if (i >= k.n) i = k.n - 1;
    return fma(k.bw, ns_norminv(u32_to_p(w.z)), k.data[i]);
}

// length of aligned segment s in epoch e (S:1285-1296,1309); returns false if no valid draw
__device__ inline bool seg_length(const DevModel &m, const ns_params &prm, const ns_key &key, uint32_t s,
                                  uint32_t epoch, int64_t &out) {
    for (uint32_t j = 0; j < NS_KDE_RETRY; ++j) {
        u32x4 w = ns_draw(key, ST_REFLEN, s, epoch, j, 0);
        double x;
        if (!prm.use_lognormal) x = kde_sample(m.kde[NS_KDE_ALIGNED], w);
        else if (prm.kind == NS_KIND_PERFECT)                                                       // S:1286-1287
            x = ns_exp(fma(prm.sd_len, ns_norminv(u32_to_p(w.z)), ns_log(prm.median_len)));
        else {                                                                                      // S:1293-1295
            u32x4 w2 = ns_draw(key, ST_REFLEN, s, epoch, j, 1);
            double tot = ns_exp(fma(prm.sd_len, ns_norminv(u32_to_p(w.z)),
                                    ns_log(prm.median_len + prm.sd_len * prm.sd_len / 2)));
            double rem = ns_pow10m1(kde_sample(m.kde[NS_KDE_HT], w2));
            if (rem < 0) continue;
            x = tot - rem;
        }
        bool keep = (prm.kind == NS_KIND_PERFECT) ? ((double)prm.min_len <= x && x <= (double)prm.max_len)
                                                  : (0 < x && x <= (double)prm.max_len);
        if (keep) { out = (int64_t)x; return true; }
    }
    return false;
}
__device__ inline int64_t gap_length(const DevModel &m, const ns_key &key, uint32_t g, uint32_t epoch) {   // S:1298-1299
    u32x4 w = ns_draw(key, ST_GAPLEN, g, epoch, 0, 0);
    int64_t gi = (int64_t)ns_pow10m1(kde_sample(m.kde[NS_KDE_GAP], w));
    return gi < 0 ? 0 : gi;
}
__device__ inline int64_t unaligned_length(const DevModel &m, const ns_params &prm, const ns_key &key, uint32_t a) {
    u32x4 w = ns_draw(key, ST_ULEN, 0, a, 0, 0);                                                    // S:1494-1495,1499
    double x = prm.use_lognormal ? ns_exp(fma(prm.sd_len, ns_norminv(u32_to_p(w.z)), ns_log(prm.median_len)))
                                 : kde_sample(m.kde[NS_KDE_UNALIGNED], w);
    return (int64_t)x;
}

// extract_read, genome branches (S:1750-1781)
__device__ inline bool extract_pos(const DevRef &ref, int64_t length, const ns_key &key, uint32_t seg, uint32_t attempt,
                                   uint32_t &chrom, uint64_t &pos) {
    uint64_t genome_len = ref.chrom_off[ref.nchrom];
    for (uint32_t j = 0; j < NS_POS_RETRY; ++j) {
        u32x4 w = ns_draw(key, ST_POS, seg, attempt, j, 0);
        uint64_t ref_pos = (uint64_t)(u53_to_p(w.x, w.y) * (double)(genome_len + 1));
        if (ref_pos > genome_len) ref_pos = genome_len;
        if (ref.circular[0]) { chrom = 0; pos = ref_pos; return true; }
        if (length > 0 && ref.nchrom > 8) {
            // the walk of S:1767-1780 ends at the chromosome that holds ref_pos: found by bisection (many-contig references)
            uint32_t lo = 0, hi = ref.nchrom;                                  // largest c with chrom_off[c] <= ref_pos
            while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (ref.chrom_off[mid] <= ref_pos) lo = mid; else hi = mid; }
            const uint64_t local = ref_pos - ref.chrom_off[lo], cl = ref.chrom_off[lo + 1] - ref.chrom_off[lo];
            if (local + (uint64_t)length <= cl) { chrom = lo; pos = local; return true; }
            continue;                                                          // does not fit (or ref_pos == genome_len): redraw
        }
        for (uint32_t c = 0; c < ref.nchrom; ++c) {
            uint64_t cl = ref.chrom_off[c + 1] - ref.chrom_off[c];
            if (ref_pos + (uint64_t)length <= cl) { chrom = c; pos = ref_pos; return true; }
            else if (ref_pos < cl) break;
            else ref_pos -= cl;
        }
    }
    return false;
}

// ---- transcriptome (S:1043-1263 without intron retention) ---------------------------------------------------
// random.choices(ecdf_length_list, weights) (S:1084): bisect_right over the running sum of the weights, clipped to n - 1
__device__ inline uint32_t trx_pick(const DevTrx &tx, double u) {
    const double v = u * tx.expr_cum[tx.n_expr - 1];
    uint32_t lo = 0, hi = tx.n_expr - 1;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (v < tx.expr_cum[mid]) hi = mid; else lo = mid + 1; }
    return lo;
}
// select_nearest_kde2d (S:108-111) on a fresh, large sample of the 2-D KDE == a draw of the aligned length from the KDE conditioned
// on the transcript length L: training point i with probability ~ exp(-(L - x_i)^2 / 2h^2), y = y_i + h N(0,1), int().  Rejection
// sampling inside |x_i - L| <= 5h; no training point there: the nearest one.  Draws: (ST_REFLEN, seg 0, attempt, idx = try, sub).
__device__ inline int64_t kde2d_cond(const DevModel &m, double L, const ns_key &key, uint32_t attempt, uint32_t sub) {
    const double *__restrict__ x = m.kde2d_x, *__restrict__ y = m.kde2d_y;
    const double h = m.kde2d_bw;
    const uint64_t n = m.kde2d_n;
    uint64_t lo, hi;
    { uint64_t a = 0, b = n; const double v = L - 5.0 * h; while (a < b) { const uint64_t md = (a + b) >> 1; if (x[md] < v) a = md + 1; else b = md; } lo = a; }
    { uint64_t a = lo, b = n; const double v = L + 5.0 * h; while (a < b) { const uint64_t md = (a + b) >> 1; if (x[md] <= v) a = md + 1; else b = md; } hi = a; }
    if (hi > lo) {
        for (uint32_t j = 0; j < NS_KDE_RETRY; ++j) {
            const u32x4 w = ns_draw(key, ST_REFLEN, 0, attempt, j, sub);
            uint64_t i = lo + (uint64_t)(u53_to_p(w.x, w.y) * (double)(hi - lo));
            if (i >= hi) i = hi - 1;
            const double dd = (L - x[i]) / h;
            if (u32_to_p(w.z) <= ns_exp(-0.5 * dd * dd)) return (int64_t)fma(h, ns_norminv(u32_to_p(w.w)), y[i]);
        }
    }
    uint64_t a = 0, b = n;
    while (a < b) { const uint64_t md = (a + b) >> 1; if (x[md] < L) a = md + 1; else b = md; }
    uint64_t i = a >= n ? n - 1 : a;
    if (a > 0 && a < n && L - x[a - 1] <= x[a] - L) i = a - 1;
    const u32x4 w = ns_draw(key, ST_REFLEN, 0, attempt, NS_KDE_RETRY, sub);
    return (int64_t)fma(h, ns_norminv(u32_to_p(w.w)), y[i]);
}
// extract_read("transcriptome", length) (S:1695-1703): a uniformly drawn transcript that is longer than the read, uniform start
__device__ inline bool extract_pos_trx_any(const DevRef &ref, int64_t length, const ns_key &key, uint32_t seg, uint32_t attempt,
                                           uint32_t &chrom, uint64_t &pos) {
    for (uint32_t j = 0; j < NS_POS_RETRY; ++j) {
        const u32x4 w = ns_draw(key, ST_POS, seg, attempt, j, 0);
        const uint32_t c = (uint32_t)(((uint64_t)w.x * ref.nchrom) >> 32);
        const uint64_t cl = ref.chrom_off[c + 1] - ref.chrom_off[c];
        if ((uint64_t)length < cl) {
            const uint64_t span = cl - (uint64_t)length + 1;
            uint64_t rp = (uint64_t)(u53_to_p(w.y, w.z) * (double)span);
            if (rp >= span) rp = span - 1;
            chrom = c; pos = rp;
            return true;
        }
    }
    return false;
}

// extract_read, metagenome branch (S:1704-1749).  species < 0: any species (gaps, unaligned reads).  Draw layout: block idx 0
// of (ST_POS, seg, attempt): word 0 species, word 1 chromosome, word 2 fall-back choice; block idx 1: 53-bit position.
__device__ inline bool extract_pos_meta(const DevRef &ref, const uint32_t *__restrict__ sp_off, uint32_t nspecies, int64_t length,
                                        int species, const ns_key &key, uint32_t seg, uint32_t attempt, uint32_t &chrom, uint64_t &pos) {
    const u32x4 wa = ns_draw(key, ST_POS, seg, attempt, 0, 0), wb = ns_draw(key, ST_POS, seg, attempt, 1, 0);
    const uint32_t s = species < 0 ? (uint32_t)(((uint64_t)wa.x * nspecies) >> 32) : (uint32_t)species;
    const uint32_t nch = sp_off[s + 1] - sp_off[s];
    uint32_t c = sp_off[s] + (uint32_t)(((uint64_t)wa.y * nch) >> 32);
    uint64_t clen = ref.chrom_off[c + 1] - ref.chrom_off[c];
    if ((uint64_t)length > clen) {                                   // S:1711-1735: a longer chromosome, of this species if any
        uint32_t nt = 0, no = 0;
        const uint32_t total = sp_off[nspecies];
        for (uint32_t ts = 0; ts < nspecies; ++ts)
            for (uint32_t k = sp_off[ts]; k < sp_off[ts + 1]; ++k)
                if ((uint64_t)length < ref.chrom_off[k + 1] - ref.chrom_off[k]) { if (ts == s) ++nt; else ++no; }
        if (!nt && !no) return false;
        const bool same = nt > 0;
        uint32_t pick = (uint32_t)(((uint64_t)wa.z * (same ? nt : no)) >> 32), seen = 0;
        for (uint32_t ts = 0, done = 0; ts < nspecies && !done; ++ts)
            for (uint32_t k = sp_off[ts]; k < sp_off[ts + 1]; ++k)
                if ((uint64_t)length < ref.chrom_off[k + 1] - ref.chrom_off[k] && ((ts == s) == same)) {
                    if (seen == pick) { c = k; done = 1; break; }
                    ++seen;
                }
        (void)total;
        clen = ref.chrom_off[c + 1] - ref.chrom_off[c];
    }
    const uint64_t span = ref.circular[c] ? clen + 1 : clen - (uint64_t)length + 1;     // randint(0, len) / randint(0, len - length)
    uint64_t rp = (uint64_t)(u53_to_p(wb.x, wb.y) * (double)span);
    if (rp >= span) rp = span - 1;
    chrom = c; pos = rp;
    return true;
}

// ---- letters ---------------------------------------------------------------------------------------------
NS_DEV uint8_t bases_atcg(uint32_t j) { return (uint8_t)(0x47435441u >> (8 * j)); }   // BASES, S:49
NS_DEV int base_rank(uint32_t c) { return c == 'A' ? 0 : c == 'T' ? 1 : c == 'C' ? 2 : c == 'G' ? 3 : -1; }
NS_DEV bool is_acgt(uint32_t c) { return ((1u << ((c - 65u) & 31u)) & 0x00080045u) != 0 && (c - 65u) < 26u; }

// case_convert (S:743-755): members in the reference's list order, packed little-endian; count in the top byte index
NS_DEV uint32_t iupac_members(uint32_t c, uint32_t &n) {
    switch (c) {
        case 'Y': n = 2; return 'C' | 'T' << 8;
        case 'R': n = 2; return 'A' | 'G' << 8;
        case 'W': n = 2; return 'A' | 'T' << 8;
        case 'S': n = 2; return 'G' | 'C' << 8;
        case 'K': n = 2; return 'T' | 'G' << 8;
        case 'M': n = 2; return 'C' | 'A' << 8;
        case 'D': n = 3; return 'A' | 'G' << 8 | 'T' << 16;
        case 'V': n = 3; return 'A' | 'C' << 8 | 'G' << 16;
        case 'H': n = 3; return 'A' | 'C' << 8 | 'T' << 16;
        case 'B': n = 3; return 'C' | 'G' << 8 | 'T' << 16;
        case 'N': case 'X': n = 4; return 0x47435441u;
        default: n = 0; return 0;
    }
}
// device form of the reference: upper-case ASCII; IUPAC ambiguity codes (and anything else, as N) carry bit 7
// so that "needs case_convert's random choice" is one AND per 4 bases
NS_DEV uint8_t normalise_base(uint32_t c) {
    c &= 0x7fu;
    if (c >= 'a' && c <= 'z') c -= 32;
    uint32_t n;
    if (is_acgt(c)) return (uint8_t)c;
    iupac_members(c, n);
    return (uint8_t)((n ? c : (uint32_t)'N') | 0x80u);
}
NS_DEV uint8_t resolve_base(uint32_t c, const ns_key &key, uint32_t seg, uint32_t attempt, uint32_t x) {
    if (!(c & 0x80u)) return (uint8_t)c;
    c &= 0x7fu;
    uint32_t n, mem = iupac_members(c, n);
    if (!n) return (uint8_t)c;
    u32x4 w = ns_draw(key, ST_IUPAC, seg, attempt, x >> 2, 0);
    uint32_t j = (uint32_t)(((uint64_t)ns_word(w, x & 3) * n) >> 32);
    return (uint8_t)(mem >> (8 * j));
}
// Payload letters of event j of a piece: one 32-bit word per event —
//   word(j, 0) = Philox(ST_SUB, seg, attempt, idx = j>>2).w[j&3]           (letters 0..15)
//   word(j, c) = Philox(ST_INS, seg, attempt, idx = j, sub = c>>2).w[c&3]  (letters 16c..16c+15, c >= 1)
// insertion letter i = 2-bit field (i&15); substitution letter i = (i&15)-th base-3 digit of the word as a fraction.
NS_DEV uint32_t payload_word(const ns_key &key, uint32_t seg, uint32_t attempt, uint32_t j, uint32_t c) {
    if (c == 0) { u32x4 w = ns_draw(key, ST_SUB, seg, attempt, j >> 2, 0); return ns_word(w, j & 3); }
    u32x4 w = ns_draw(key, ST_INS, seg, attempt, j, c >> 2);
    return ns_word(w, c & 3);
}
NS_DEV uint8_t mis_from_digit(uint32_t cur, uint32_t dg) {      // S:1968-1972
    int rc = base_rank(cur);
    uint32_t rk = dg + ((int)dg >= rc ? 1u : 0u);
    if (rc < 0) rk = dg;
    return bases_atcg(rk);
}
NS_DEV uint32_t next_digit3(uint32_t &frac) {
    uint64_t p = (uint64_t)frac * 3u;
    frac = (uint32_t)p;
    return (uint32_t)(p >> 32);
}
__device__ __forceinline__ uint8_t mis_letter(uint32_t cur, const ns_key &key, uint32_t seg, uint32_t attempt, uint32_t j, uint32_t i) {
    uint32_t frac = payload_word(key, seg, attempt, j, i >> 4), dg = 0;
    for (uint32_t t = 0; t <= (i & 15); ++t) dg = next_digit3(frac);
    return mis_from_digit(cur, dg);
}
__device__ __forceinline__ uint8_t ins_letter(const ns_key &key, uint32_t seg, uint32_t attempt, uint32_t j, uint32_t i) {   // S:1990
    return bases_atcg((payload_word(key, seg, attempt, j, i >> 4) >> (2 * (i & 15))) & 3u);
}
__device__ __forceinline__ uint8_t ht_letter(const ns_key &key, uint32_t stream, uint32_t attempt, uint32_t i) {             // S:1426-1427
    u32x4 w = ns_draw(key, stream, 0, attempt, i >> 6, 0);
    return bases_atcg((ns_word(w, (i >> 4) & 3) >> (2 * (i & 15))) & 3u);
}
// the same count through a 1024-bucket look-up table: the count at the start of the bucket in bits 7.., and below it 128 minus the
// offset of the only threshold inside the bucket — adding h & 63 carries into the count exactly when h is at or above that
// threshold (buckets are 64 wide; a bucket with several thresholds is walked: the loader's tables have none, model.py)
__device__ __forceinline__ uint8_t qual_value_lut(const uint32_t *__restrict__ thr, const uint16_t *__restrict__ lut, uint32_t h) {
    const uint32_t e = lut[h >> 6];
    if (!(e & 0x8000u)) return (uint8_t)(((e & 0x7fffu) + (h & 63u)) >> 7);
    uint32_t q = (e >> 7) & 0x7fu;
    while (q < NS_QUAL_LEVELS - 1 && h >= thr[q]) ++q;
    return (uint8_t)q;
}
__device__ __forceinline__ uint8_t qual_value(const uint32_t *__restrict__ thr, uint32_t h) {
    // q = #{j in [0,126] : h >= thr[j]}; thr is non-decreasing -> binary search for the first thr[j] > h
    uint32_t lo = 0, hi = NS_QUAL_LEVELS - 1;
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if (h >= thr[mid]) lo = mid + 1; else hi = mid;
    }
    return (uint8_t)lo;
}
__device__ __forceinline__ uint8_t qual_at(const DevModel &m, int cls, const ns_key &key, uint32_t stream, uint32_t seg,
                                           uint32_t attempt, uint32_t mpos) {
    u32x4 w = ns_draw(key, stream, seg, attempt, mpos >> 3, 0);
    uint32_t h = (ns_word(w, (mpos & 7) >> 1) >> (16 * (mpos & 1))) & 0xffffu;
    return qual_value(m.qual_thr + cls * NS_QUAL_LEVELS, h);
}

// (32-bit values — positions, run lengths, lengths — take the 32-bit forms: a division of a 64-bit value by ten is a multi-instruction
// sequence on this target, and k_errlen / k_errlog / the names format millions of numbers per batch)
NS_DEV uint32_t dec_digits(uint32_t v) {
    return v < 10u ? 1u : v < 100u ? 2u : v < 1000u ? 3u : v < 10000u ? 4u : v < 100000u ? 5u : v < 1000000u ? 6u : v < 10000000u ? 7u :
           v < 100000000u ? 8u : v < 1000000000u ? 9u : 10u;
}
NS_DEV uint8_t *put_dec(uint8_t *p, uint32_t v) {
    const uint32_t n = dec_digits(v);
    for (uint32_t i = 0; i < n; ++i) { const uint32_t q = v / 10u; p[n - 1 - i] = (uint8_t)('0' + (v - 10u * q)); v = q; }
    return p + n;
}
// the same through any pointer type (k_errlog: volatile LDS bytes — every digit stays ONE byte store; a merged store at an odd LDS
// address costs eight byte stores)
template <class P>
NS_DEV P put_dec_p(P p, uint32_t v) {
    const uint32_t n = dec_digits(v);
    for (uint32_t i = 0; i < n; ++i) { const uint32_t q = v / 10u; p[n - 1 - i] = (uint8_t)('0' + (v - 10u * q)); v = q; }
    return p + n;
}
__device__ __forceinline__ uint32_t dec_digits(uint64_t v) {
    uint32_t n = 1;
    while (v >= 10) { v /= 10; ++n; }
    return n;
}
__device__ __forceinline__ uint8_t *put_dec(uint8_t *p, uint64_t v) {
    uint32_t n = dec_digits(v);
    for (uint32_t i = 0; i < n; ++i) { p[n - 1 - i] = (uint8_t)('0' + v % 10); v /= 10; }
    return p + n;
}
__device__ __forceinline__ uint8_t complement(uint32_t x) {        // reverse_complement, S:1675-1680
    return x == 'A' ? 'T' : x == 'T' ? 'A' : x == 'C' ? 'G' : x == 'G' ? 'C' : (uint8_t)x;
}
