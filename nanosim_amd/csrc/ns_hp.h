// ns_hp.h — -k/--KmerBias: the homopolymer filter of mutate_read (S:1920-1947) and mutate_homo (S:618-705).
//
// Lengths change AFTER the mutated segment exists, so the mode has two record passes:
//   k_hp_filter_w                     wave/read  drop the events that overlap a homopolymer of the un-mutated segment, re-pack the rest
//   k_materialise<., MAT_HP_SCRATCH>  wave/read  the pieces of the read before mutate_homo, forward strand, into the scratch buffer
//                                                (FASTQ: every base carries its quality class in bits 3 / 5)
//   k_hp_scan, k_hp_drain             wave/read  runs >= k of the scratch segments -> new run lengths, mismatches -> an edit list per
//                                                piece; final lengths, final length check (S:1429): a failing read bumps its attempt
//                                                state and the batch is re-run
//   k_materialise<., MAT_HP_FINAL>    wave/read  scratch + edit list -> the record (qualities drawn here, strand, T -> U)
#pragma once
#include "ns_materialise.h"

__device__ __forceinline__ uint8_t conv_base(const DevRef &ref, const PieceCtx &pc, const ns_key &key, uint32_t a, uint32_t x) {
    return resolve_base(ref_base_at(ref, pc, x), key, pc.sid, a, x);       // case_convert()ed base x of the segment
}
// is base x of the converted segment inside a run of >= k identical bases?
__device__ inline bool in_hp_run(const DevRef &ref, const PieceCtx &pc, const ns_key &key, uint32_t a, int64_t x, int64_t k) {
    if (x < 0 || x >= (int64_t)pc.ref_len) return false;
    const uint8_t b = conv_base(ref, pc, key, a, (uint32_t)x);
    int64_t s = x, e = x + 1;
    while (s > 0 && e - s < k && conv_base(ref, pc, key, a, (uint32_t)(s - 1)) == b) --s;
    while (e < (int64_t)pc.ref_len && e - s < k && conv_base(ref, pc, key, a, (uint32_t)e) == b) ++e;
    return e - s >= k;
}

// 0x80 in every byte of x that is not zero
__device__ __forceinline__ uint32_t nonzero_bytes(uint32_t x) { return (((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u; }
__device__ __forceinline__ uint32_t movemask4(uint32_t flags80) { return (((flags80 >> 7) * 0x00204081u) >> 21) & 0xfu; }

// The same for k <= 16 on a segment that does not wrap the origin of a circular chromosome: the 32 bases around x come in two
// 16-byte loads (the engine's copy of the reference and the splice arena are padded), "differs from base x" becomes a bit per base,
// and the run around x is two bit scans — instead of up to 2k dependent byte loads.  An IUPAC code in the window: the generic walk.
__device__ inline bool in_hp_run_win(const DevRef &ref, const PieceCtx &pc, const ns_key &key, uint32_t a, int64_t x, int64_t k) {
    if (x < 0 || x >= (int64_t)pc.ref_len) return false;
    const uint8_t *p = ref.bases + pc.chrom_base + pc.pos + x - 15;       // window = segment positions [x - 15, x + 16]
    uint4 lo, hi;
    __builtin_memcpy(&lo, p, 16); __builtin_memcpy(&hi, p + 16, 16);
    if ((lo.x | lo.y | lo.z | lo.w | hi.x | hi.y | hi.z | hi.w) & 0x80808080u) return in_hp_run(ref, pc, key, a, x, k);
    const uint32_t bb = (lo.w >> 24) * 0x01010101u;                        // base x in every byte
    uint32_t ne = movemask4(nonzero_bytes(lo.x ^ bb)) | movemask4(nonzero_bytes(lo.y ^ bb)) << 4 | movemask4(nonzero_bytes(lo.z ^ bb)) << 8 |
                  movemask4(nonzero_bytes(lo.w ^ bb)) << 12 | movemask4(nonzero_bytes(hi.x ^ bb)) << 16 | movemask4(nonzero_bytes(hi.y ^ bb)) << 20 |
                  movemask4(nonzero_bytes(hi.z ^ bb)) << 24 | movemask4(nonzero_bytes(hi.w ^ bb)) << 28;
    if (x < 15) ne |= (1u << (15 - x)) - 1u;                               // positions outside the segment end the run
    const int64_t top = (int64_t)pc.ref_len - x + 14;                      // highest window bit inside the segment
    if (top < 31) ne |= 0xffffffffu << (top + 1);
    const uint32_t up = ne >> 15, dn = ne << 17;
    const uint32_t r = up ? (uint32_t)__builtin_ctz(up) : 17u;            // base x and the equal bases behind it
    const uint32_t l = dn ? (uint32_t)__builtin_clz(dn) : 15u;            // equal bases in front of it
    return (int64_t)(l + r) >= k;
}

// the same for k <= 8 with ONE 16-byte load: window = segment positions [x - 7, x + 8] (7 bases in front of x and 8 behind it decide
// any run of up to 8; a run that leaves the window is long enough anyway)
__device__ inline bool in_hp_run_win8(const DevRef &ref, const PieceCtx &pc, const ns_key &key, uint32_t a, int64_t x, int64_t k) {
    if (x < 0 || x >= (int64_t)pc.ref_len) return false;
    const uint8_t *p = ref.bases + pc.chrom_base + pc.pos + x - 7;
    uint4 v;
    __builtin_memcpy(&v, p, 16);
    if ((v.x | v.y | v.z | v.w) & 0x80808080u) return in_hp_run(ref, pc, key, a, x, k);
    const uint32_t bb = (v.y >> 24) * 0x01010101u;                         // base x (byte 7) in every byte
    uint32_t ne = movemask4(nonzero_bytes(v.x ^ bb)) | movemask4(nonzero_bytes(v.y ^ bb)) << 4 | movemask4(nonzero_bytes(v.z ^ bb)) << 8 |
                  movemask4(nonzero_bytes(v.w ^ bb)) << 12;
    if (x < 7) ne |= (1u << (7 - x)) - 1u;                                 // positions outside the segment end the run
    const int64_t top = (int64_t)pc.ref_len - x + 6;                       // highest window bit inside the segment
    if (top < 15) ne |= 0xffffu << (top + 1);
    const uint32_t up = (ne & 0xffffu) >> 7, dn = ne << 25;
    const uint32_t r = up ? (uint32_t)__builtin_ctz(up) : 9u;             // base x and the equal bases behind it
    const uint32_t l = dn ? (uint32_t)__builtin_clz(dn) : 7u;             // equal bases in front of it
    return (int64_t)(l + r) >= k;
}

// get_nd_par (src/model_homopolymer_lengths.py:246-260).  (The two parameter rows are read with static indices and selected field by
// field: a dynamic index into the kernel argument would make the compiler keep a copy of the table in scratch memory.)
__device__ __forceinline__ void hp_nd_par(const DevModel &m, uint32_t base, uint32_t len, double &mu, double &sigma) {
    const bool at = base == 'A' || base == 'T';
    const ns_hp_class &h0 = m.hp[0], &h1 = m.hp[1];
    const double x = (double)len;
    double y = (at ? h0.konst : h1.konst) + (at ? h0.alpha1 : h1.alpha1) * x;
    const uint32_t nb = at ? h0.n_breaks : h1.n_breaks;
#pragma unroll
    for (uint32_t j = 0; j < NS_HP_MAX_BREAKS; ++j) {
        if (j < nb) {
            const double d = x - (at ? h0.breakpoint[j] : h1.breakpoint[j]);
            y += (at ? h0.beta[j] : h1.beta[j]) * (d > 0 ? d : 0.0);
        }
    }
    mu = y; sigma = (at ? h0.intercept : h1.intercept) + (at ? h0.slope : h1.slope) * x;
}
// new length of the run [s, s+L) of base b (S:644-654, 665)
__device__ __forceinline__ uint32_t hp_new_size(const DevModel &m, const ns_key &key, uint32_t sid, uint32_t a, uint32_t s,
                                                uint32_t L, uint32_t b) {
    double mu, sigma;
    hp_nd_par(m, b, L, mu, sigma);
    u32x4 w = ns_draw(key, ST_HPLEN, sid, a, s, 0);
    double x = fma(sigma, ns_norminv(u32_to_p(w.x)), mu);
    if (x < 0) x = 0;
    return (uint32_t)(int64_t)rint(x);
}
// The events that turn the run [s0, s0 + L) of `base` into its re-sampled form (mutate_homo, S:657-700), in ascending position,
// handed to put(pos, type, len, word); returns their number.  New base x of the run (0 <= x < size):
//   mismatch (S:671-673) iff u32_to_p(Philox(ST_HPMIS, sid, a, idx = s0, sub = x >> 2).word[x & 3]) <= hp_mis_rate;
//   its letter (S:674-679): choice word c = Philox(ST_HPMIS, sid, a, idx = s0, sub = 0x80000000 | x).word[0], j = (c' * 3) >> 32 picks
//   among the three other bases in "ATCG" order (kept base: c' = c with bit 0 replaced by the first-mismatch flag; appended base: c' = c).
// A contraction keeps the run's LAST `size` bases (S:688-690: the first |diff| qualities go): one deletion at s0.  Appended bases
// (S:692-695, class 'ins') are insertions of <= 15 letters at s0 + L, 2 bits per letter; the run's first mismatch takes the 'mis' class
// (S:697-700): bit 0 of a substitution's word, bit 31 of an insertion's word (the letter is then letter 0 of its insertion).
template <typename F>
__device__ __forceinline__ uint32_t hp_run_events(double hp_mis_rate, const ns_key &key, uint32_t sid, uint32_t a, uint32_t s0, uint32_t L,
                                                  uint32_t size, uint32_t base, F &&put) {
    uint32_t n = 0;
    u32x4 w{0, 0, 0, 0}; uint32_t wblk = 0xffffffffu;
    auto is_mis = [&](uint32_t x) {
        if ((x >> 2) != wblk) { wblk = x >> 2; w = ns_draw(key, ST_HPMIS, sid, a, s0, wblk); }
        return u32_to_p(ns_word(w, x & 3u)) <= hp_mis_rate;
    };
    auto choice = [&](uint32_t x) { return ns_draw(key, ST_HPMIS, sid, a, s0, 0x80000000u | x).x; };
    bool first = true;
    const uint32_t kept = min(size, L), off = L - kept;
    for (uint32_t d = L - kept, pos = s0; d;) { const uint32_t c = min(d, NS_EV_LEN_MAX); put(pos, (uint32_t)NS_DEL, c, 0u); pos += c; d -= c; ++n; }
    // the kept bases four at a time (one Philox block; p <= rate <=> u < ns_thr_gt(rate)): mismatches are rare, the loop over the
    // bases of the wavefront's longest run is what the drain spends its time on
    const uint64_t thr = ns_thr_gt(hp_mis_rate);
    for (uint32_t x0 = 0; x0 < kept; x0 += 4) {
        wblk = x0 >> 2; w = ns_draw(key, ST_HPMIS, sid, a, s0, wblk);
        uint32_t m4 = ((uint64_t)w.x < thr ? 1u : 0u) | ((uint64_t)w.y < thr ? 2u : 0u) | ((uint64_t)w.z < thr ? 4u : 0u) | ((uint64_t)w.w < thr ? 8u : 0u);
        if (kept - x0 < 4u) m4 &= (1u << (kept - x0)) - 1u;
        for (; m4; m4 &= m4 - 1u) {
            const uint32_t x = x0 + (uint32_t)__builtin_ctz(m4);
            put(s0 + off + x, (uint32_t)NS_MIS, 1u, (choice(x) & ~1u) | (first ? 1u : 0u)); first = false; ++n;
        }
    }
    const uint32_t code_b = (uint32_t)base_rank(base);
    for (uint32_t x = L; x < size;) {
        uint32_t word = 0, cnt = 0; bool flag = false;
        while (x < size && cnt < 15u) {
            const bool mm = is_mis(x);
            if (mm && first && cnt) break;                     // the run's first mismatch opens its own insertion
            uint32_t code = code_b;
            if (mm) {
                const uint32_t j = (uint32_t)(((uint64_t)choice(x) * 3u) >> 32);
                code = j + (j >= code_b ? 1u : 0u);
                if (first) { flag = true; first = false; }
            }
            word |= code << (2u * cnt); ++cnt; ++x;
        }
        put(s0 + L, (uint32_t)NS_INS, cnt, word | (flag ? 0x80000000u : 0u)); ++n;
    }
    return n;
}

