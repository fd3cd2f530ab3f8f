// ns_pack.h — host side of ns_load_model: the tables of the event chains packed into ONE blob of 8-byte words (ns_chain.h reads it;
// k_chain copies its first n_words_lds words to LDS; the wave-per-read unaligned chain its first n_words_mix).  Plain host code without a HIP call, so that the CPU test suite can pack a model and
// run the chain source compiled for the host against the oracle (tests/chain_host.hip).
//
// The image (offsets in ChainTab, ns_device.h).  LDS part:
//   trans      21 thresholds: rows start, mis, ins, del, mis0, ins0, del0 x (p < a, p < a + b, -)                       (S:1860-1864)
//   mix_w      3 thresholds: tmp_rand < weight                                                                          (mm:44, 54)
//   6 x        the inverse-CDF table of (error type, mixture component) as thresholds of p > cdf[v], each followed by its guide: 33 bytes,
//              g[l] = thresholds at or below the smallest draw with l leading one bits (the walk of run_length_r starts there)
//   mix_rec    6 records {table offset | thresholds << 32, guide offset}, index 2 * type + component — ONE LDS read instead of three
//              per-lane indexed kernel-argument loads
//   fm_guide, mm_guide   256-entry guides of the ECDF columns: g[i] = #{s : hi[s] < i/256}
//   mm_bin, mm_bin_lut, mm_seg_off, mm_vlo0   bins of the previous match length (S:1891-1893), the bin of a length < 256, the columns' ranges
//   mm_gv      (LDS: the hot prefix of every column, ns_device.h; the full columns mm_gv_full and the first-match column fm_gv lie in the global part)
//              per ECDF segment ONE word: bits 0..32 the threshold ns_thr_gt(hi[s]) (p > hi[s] <=> u >= thr, exactly); bit 33: the
//              segment is one unit wide and every draw inside it gives vlo — bits 35.. = vhi; bit 34: 1..15 units wide with steps inside —
//              bits 35.. = index into sub2 of {vhi | steps << 32} followed by the step thresholds; neither bit: the fp64 formula on the
//              global tables
//   pm_lut     previous match v < 256 -> {first segment of its column | segments << 32 | bin << 56}
//   sub2       the step lists of the narrow segments
// Global part (wide segments, the cooperative chain, models whose image does not fit LDS): fm_hi, mm_hi, fm_vhi, mm_vhi as fp64.
#pragma once
#include <math.h>
#define NS_PACK_THR(g) ((g) & 0x1ffffffffull)        // the threshold field of a segment word (NS_G_THR of ns_chain.h)
#include <string.h>
#include <vector>
#include "ns_device.h"

// nseg: number of segments of all match-length columns together (t->mm_seg_off[t->mm_nbins]).  whole: every value edge is a whole number
// that fits its field and the step lists fit their index field — the condition for the integer LDS image (k_chain<LDS>).
static inline void ns_pack_chain_tables(const ns_model_tables *t, uint32_t nseg, ChainTab &ct, std::vector<uint64_t> &blob, bool &whole,
                                        uint32_t force_tail_bits = 0) {
    blob.clear();
    auto put_d = [&](const double *src, size_t n) { uint32_t off = (uint32_t)blob.size(); blob.resize(off + n);
                                                     memcpy(blob.data() + off, src, n * 8); return off; };
    auto put_raw = [&](const void *src, size_t bytes) { uint32_t off = (uint32_t)blob.size(); blob.resize(off + (bytes + 7) / 8, 0);
                                                         memcpy(blob.data() + off, src, bytes); return off; };
    auto put_q = [&](const std::vector<uint64_t> &v) { uint32_t off = (uint32_t)blob.size(); blob.insert(blob.end(), v.begin(), v.end()); return off; };
    auto guide = [&](const double *hi, uint32_t n) {          // g[i] = #{s : hi[s] < i/256}: lower bound of the segment of any p >= i/256
        std::vector<uint16_t> g(256);
        uint32_t sidx = 0;
        for (uint32_t i = 0; i < 256; ++i) {
            double edge = (double)i / 256.0;
            while (sidx < n && hi[sidx] < edge) ++sidx;
            g[i] = (uint16_t)(sidx > 65535u ? 65535u : sidx);
        }
        return g;
    };
    // the tables that are only ever COMPARED with a draw are stored as integer thresholds of the 32-bit draw (ns_thr_lt / ns_thr_gt)
    auto put_thr = [&](const double *src, size_t n, bool gt) {
        uint32_t off = (uint32_t)blob.size(); blob.resize(off + n);
        for (size_t i = 0; i < n; ++i) blob[off + i] = gt ? ns_thr_gt(src[i]) : ns_thr_lt(src[i]);
        return off; };
    whole = true;
    ct.trans = put_thr(&t->trans[0][0], 21, false);          // p < a, p < a + b            (S:1860-1864)
    ct.mix_w = put_thr(t->mix_w, 3, false);                   // tmp_rand < weight           (mm:44, 54)
    uint64_t rec[12];
    for (int ty = 0; ty < 3; ++ty)                            // p > cdf[v]: walk of the inverse-CDF tables
        for (int c = 0; c < 2; ++c) {
            const uint32_t nn = t->mix_n[ty][c];
            const uint32_t cdf = put_thr(t->mix_cdf[ty][c], nn, true);
            // guide of the walk by the number of leading one bits of the draw (the tail of a run-length CDF is geometric: a constant
            // number of thresholds per halving of 1 - p): g[l] = thresholds at or below the smallest draw with l leading ones — a
            // lower bound of the walk's result for every draw of that class, so the walk starts there instead of at 0
            uint8_t g2[40] = {0};
            const uint64_t *G = blob.data() + cdf;
            for (uint32_t l = 0; l <= 32; ++l) {
                const uint64_t lo_u = l == 0 ? 0ull : (0xffffffffull << (32 - l)) & 0xffffffffull;
                uint32_t v = 0;
                while (v + 1 < nn && lo_u >= G[v]) ++v;
                g2[l] = (uint8_t)(v > 255u ? 255u : v);
            }
            const uint32_t gd = put_raw(g2, 40);              // (also the word behind the table that run_length_r reads and never uses)
            rec[2 * (2 * ty + c)] = (uint64_t)cdf | (uint64_t)nn << 32;
            rec[2 * (2 * ty + c) + 1] = gd;
        }
    ct.mix_rec = put_raw(rec, sizeof rec);
    ct.n_words_mix = (uint32_t)blob.size();                   // everything unaligned_error_list reads lies in front of here
    ct.fm_n = t->fm_nseg; ct.fm_vlo0 = t->fm_vlo0;
    { auto g = guide(t->fm_hi, t->fm_nseg); ct.fm_guide = put_raw(g.data(), 512); }
    ct.mm_nbins = t->mm_nbins;
    std::vector<int32_t> bins(2 * (size_t)t->mm_nbins);
    for (uint32_t b = 0; b < t->mm_nbins; ++b) {
        auto clamp = [](int64_t v) { return (int32_t)(v > 0x7fffffff ? 0x7fffffff : v < -0x7fffffff ? -0x7fffffff : v); };
        bins[2 * b] = clamp(t->mm_bin_lo[b]); bins[2 * b + 1] = clamp(t->mm_bin_hi[b]);
    }
    ct.mm_bin = put_raw(bins.data(), bins.size() * 4);
    std::vector<uint8_t> lut(256);                            // direct bin of a previous match length < 256 (first bin with lo <= v < hi, else the
    for (int v = 0; v < 256; ++v) {                           // last bin, S:1891-1893)
        uint32_t b = 0;
        for (; b < t->mm_nbins; ++b) if (bins[2 * b] <= v && v < bins[2 * b + 1]) break;
        if (b >= t->mm_nbins) b = t->mm_nbins - 1;
        lut[v] = (uint8_t)b;
    }
    ct.mm_bin_lut = put_raw(lut.data(), 256);
    ct.mm_seg_off = put_raw(t->mm_seg_off, ((size_t)t->mm_nbins + 1) * 4);
    ct.mm_vlo0 = put_d(t->mm_vlo0, t->mm_nbins);
    std::vector<uint16_t> gall;
    for (uint32_t b = 0; b < t->mm_nbins; ++b) {
        auto g = guide(t->mm_hi + t->mm_seg_off[b], t->mm_seg_off[b + 1] - t->mm_seg_off[b]);
        gall.insert(gall.end(), g.begin(), g.end());
    }
    ct.mm_guide = put_raw(gall.data(), gall.size() * 2);
    // One word per ECDF segment.  The value edges are whole numbers in every model read_analysis.py writes (its bins are "i-(i+1)"), and
    // inside a segment (vlo, vhi] the interpolation floor((p - plo) / (hs - plo) * (vs - vlo) + vlo) of S:1847 / S:1897 is a
    // non-decreasing step function of the draw: its steps are found here with the arithmetic of the fp64 formula (this file is compiled
    // with -ffp-contract=off, like the device code and the oracle) — ecdf_lookup_gv resolves a draw without floating point.
    auto segments = [&](const double *hi, const double *src, size_t n, double vlo0, std::vector<uint64_t> &out, std::vector<uint64_t> &sub2) {
        for (size_t i = 0; i < n; ++i) {
            if (!(src[i] >= 0 && src[i] < 536870912.0 && src[i] == floor(src[i]))) whole = false;       // (29 bits behind the flags)
            const uint32_t e = (uint32_t)(src[i] < 0 ? 0 : src[i] >= 2147483647.0 ? 2147483647.0 : src[i]);
            const double hs = hi[i], plo = i ? hi[i - 1] : 0.0, vs = src[i], vlo = i ? src[i - 1] : vlo0;
            const double w = vs - vlo;
            uint64_t word = ns_thr_gt(hs);                     // p > hi[s]  <=>  u >= thr
            if (w >= 1.0 && w <= 15.0 && w == floor(w) && hs > plo && vlo == floor(vlo) && vlo >= 0) {
                auto f = [&](uint64_t u) { const double pp = u32_to_p((uint32_t)u); return floor((pp - plo) / (hs - plo) * (vs - vlo) + vlo); };
                const uint64_t u0 = i ? ns_thr_gt(plo) : 0ull, u1 = ns_thr_gt(hs);     // the draws of the segment: [u0, u1)
                uint64_t thr[15];
                for (uint32_t k = 1; k <= (uint32_t)w; ++k) {                          // smallest draw of the segment that gives >= vlo + k
                    uint64_t lo = u0, hi2 = u1;                                         // (f is non-decreasing in the draw)
                    while (lo < hi2) { const uint64_t mid = lo + ((hi2 - lo) >> 1); if (f(mid) >= vlo + (double)k) hi2 = mid; else lo = mid + 1; }
                    thr[k - 1] = lo >= u1 ? (1ull << 32) : lo;
                }
                if (w == 1.0 && thr[0] == (1ull << 32)) word |= 1ull << 33 | (uint64_t)e << 35;      // one unit wide, every draw gives vlo
                else {
                    word |= 1ull << 34 | (uint64_t)sub2.size() << 35;
                    sub2.push_back((uint64_t)e | (uint64_t)(uint32_t)w << 32);
                    sub2.insert(sub2.end(), thr, thr + (uint32_t)w);
                }
            }
            out.push_back(word);
        } };
    // ---- the FULL columns (global memory) ----
    std::vector<uint64_t> fm_gv, mm_full, sub2_full;
    segments(t->fm_hi, t->fm_vhi, t->fm_nseg, t->fm_vlo0, fm_gv, sub2_full);
    for (uint32_t b = 0; b < t->mm_nbins; ++b) {               // per column: its first segment starts at the column's vlo0
        const uint32_t o = t->mm_seg_off[b];
        segments(t->mm_hi + o, t->mm_vhi + o, t->mm_seg_off[b + 1] - o, t->mm_vlo0[b], mm_full, sub2_full);
    }
    if (sub2_full.size() >= (1u << 28)) whole = false;
    sub2_full.push_back(0);
    std::vector<uint64_t> pm_full(256);
    for (uint32_t v = 0; v < 256; ++v) {
        const uint32_t b = lut[v], o = t->mm_seg_off[b], nc = t->mm_seg_off[b + 1] - o;
        if (nc >= (1u << 24)) whole = false;
        pm_full[v] = (uint64_t)o | (uint64_t)(nc & 0xffffffu) << 32 | (uint64_t)b << 56;
    }
    // ---- the HOT PREFIX of every match-length column (the LDS part) ----
    // A column of a trained model has as many rows as the longest match in the training data (1 500 x 15 bins x 8 bytes = 180 KB: more than a
    // CU's LDS), but a draw lands behind row r with the probability the column has left there.  The prefix of a column ends at its first
    // segment whose threshold leaves <= 2^-tail_bits; the chain's fast path only accepts a segment INSIDE the prefix (it tests s < ncol with
    // the prefix length), everything else — one draw in 2^tail_bits — goes through next_match_gv on the full column: same thresholds, same
    // answer.  tail_bits: the largest of 14 .. 12 whose image fits 24 KB (five 256-thread workgroups with their 8 KB of event staging in a CU's
    // 160 KB), else of 14 .. 8 that fits 44 KB (two 512-thread workgroups with 16 KB each: four waves per SIMD; a model of very long matches
    // gives up coverage for a place in LDS); 0 = the prefixes are the full columns (small models: the image of the bench model is 24 KB either way).
    const size_t head_words = blob.size();
    auto prefix_len = [&](uint32_t b, uint32_t bits) {
        const uint32_t o = t->mm_seg_off[b], nc = t->mm_seg_off[b + 1] - o;
        if (!bits) return nc;
        const uint64_t cut = (1ull << 32) - (1ull << (32 - bits));
        uint32_t r = 0;
        while (r < nc && NS_PACK_THR(mm_full[o + r]) < cut) ++r;    // segments whose upper edge is still below the cut ...
        return r < nc ? r + 1 : nc;                                   // ... and the one that crosses it
    };
    auto image_words = [&](uint32_t tb) {                     // (+ the prefixes' step lists: at most the full ones)
        size_t n = head_words + 256 + 64 + t->mm_nbins + sub2_full.size();
        for (uint32_t b = 0; b < t->mm_nbins; ++b) n += prefix_len(b, tb);
        return n; };
    uint32_t bits = 0;
    if (image_words(0) * 8 > 24 * 1024) {
        bits = 8;
        for (uint32_t limit : {24u * 1024u, 44u * 1024u}) {
            uint32_t best = 0;
            for (uint32_t tb = 14; tb >= (limit == 24u * 1024u ? 12u : 8u); --tb) if (image_words(tb) * 8 <= limit) { best = tb; break; }
            if (best) { bits = best; break; }
        }
    }
    if (force_tail_bits) bits = force_tail_bits;              // (NS_TAIL_BITS: A/B runs)
#ifdef NS_PACK_TAIL_BITS
    bits = NS_PACK_TAIL_BITS;                                 // test builds (tests/test_chain_host.py): short prefixes, so that the full-column path is taken often
#endif
    ct.tail_bits = bits;
    std::vector<uint64_t> mm_gv, sub2;
    std::vector<uint32_t> pre_off(t->mm_nbins + 1, 0);
    for (uint32_t b = 0; b < t->mm_nbins; ++b) {
        const uint32_t o = t->mm_seg_off[b], r = prefix_len(b, bits);
        pre_off[b] = (uint32_t)mm_gv.size();
        segments(t->mm_hi + o, t->mm_vhi + o, r, t->mm_vlo0[b], mm_gv, sub2);
    }
    pre_off[t->mm_nbins] = (uint32_t)mm_gv.size();
    std::vector<uint64_t> pm(256);
    for (uint32_t v = 0; v < 256; ++v) {
        const uint32_t b = lut[v];
        pm[v] = (uint64_t)pre_off[b] | (uint64_t)((pre_off[b + 1] - pre_off[b]) & 0xffffffu) << 32 | (uint64_t)b << 56;
    }
    // (the chain reads the word behind a column's last segment and never uses it: every table below is followed by another one)
    sub2.push_back(0);
    std::vector<uint64_t> pmb(t->mm_nbins);
    for (uint32_t b = 0; b < t->mm_nbins; ++b) pmb[b] = (uint64_t)pre_off[b] | (uint64_t)((pre_off[b + 1] - pre_off[b]) & 0xffffffu) << 32 | (uint64_t)b << 56;
    ct.mm_gv = put_q(mm_gv); ct.pm_lut = put_q(pm); ct.pm_bin = put_q(pmb); ct.sub2 = put_q(sub2);
    ct.n_words_lds = (uint32_t)blob.size();
    ct.fm_gv = put_q(fm_gv); ct.mm_gv_full = put_q(mm_full); ct.pm_full = put_q(pm_full); ct.sub2_full = put_q(sub2_full);
    {   // the 65 536-cell guides of the full columns (coop_error_list): g[i] = #{s : thr(s) <= i << 16}
        std::vector<uint16_t> g16((size_t)t->mm_nbins * 65536u);
        for (uint32_t b = 0; b < t->mm_nbins; ++b) {
            const uint32_t o = t->mm_seg_off[b], nc = t->mm_seg_off[b + 1] - o;
            uint32_t sidx = 0;
            for (uint32_t i = 0; i < 65536u; ++i) {
                const uint64_t lo_u = (uint64_t)i << 16;
                while (sidx < nc && NS_PACK_THR(mm_full[o + sidx]) <= lo_u) ++sidx;
                g16[(size_t)b * 65536u + i] = (uint16_t)(sidx > 65535u ? 65535u : sidx);
            }
        }
        ct.mm_g16 = put_raw(g16.data(), g16.size() * 2);
    }
    ct.fm_hi = put_d(t->fm_hi, t->fm_nseg); ct.mm_hi = put_d(t->mm_hi, nseg);      // fp64 tables: wide segments, cooperative chain, chain_error_list_g
    ct.fm_vhi = put_d(t->fm_vhi, t->fm_nseg); ct.mm_vhi = put_d(t->mm_vhi, nseg);
    ct.int_image = whole ? 1u : 0u;
    ct.n_words = (uint32_t)blob.size();
}
