// ns_pack.h — host side of ns_load_model: the tables of the event chains packed into ONE blob of 8-byte words (ns_chain.h reads it;
// k_chain copies its first n_words_lds words to LDS).  Plain host code without a HIP call, so that the CPU test suite can pack a model and
// run the chain source compiled for the host against the oracle (tests/chain_host.hip).
#pragma once
#include <math.h>
#include <string.h>
#include <utility>
#include <vector>
#include "ns_device.h"

// nseg: number of segments of all match-length columns together (t->mm_seg_off[t->mm_nbins]).  whole: every value edge is a whole number
// below 2^31 and the step lists fit their index field — the condition for the integer LDS image (k_chain<LDS>).
// layout (NS_CHAIN_LAYOUT; 0 in the product build): bit 0 adds the run-length records, bit 1 the one-word ECDF segments and the column
// table (ChainTab, NS_CHAIN_TABS2) — ns_pack_layout2 below rearranges the image this function builds.
#ifdef NS_CHAIN_TABS2
static inline void ns_pack_layout2(ChainTab &ct, std::vector<uint64_t> &blob, bool &whole, uint32_t layout, uint32_t fm_n, uint32_t nseg,
                                   const std::vector<std::pair<uint32_t, uint32_t>> &segs);
#endif
static inline void ns_pack_chain_tables(const ns_model_tables *t, uint32_t nseg, ChainTab &ct, std::vector<uint64_t> &blob, bool &whole,
                                        uint32_t layout = 0) {
    blob.clear();
    std::vector<std::pair<uint32_t, uint32_t>> segs;          // (first word, words) of every table, in blob order
    // ---- pack the chain tables into one blob of 8-byte words (its first part is copied to LDS by k_chain) ----
    auto put_d = [&](const double *src, size_t n) { uint32_t off = (uint32_t)blob.size(); blob.resize(off + n);
                                                     memcpy(blob.data() + off, src, n * 8); segs.emplace_back(off, (uint32_t)n); return off; };
    auto put_raw = [&](const void *src, size_t bytes) { uint32_t off = (uint32_t)blob.size(); blob.resize(off + (bytes + 7) / 8, 0);
                                                         memcpy(blob.data() + off, src, bytes); segs.emplace_back(off, (uint32_t)((bytes + 7) / 8));
                                                         return off; };
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
        segs.emplace_back(off, (uint32_t)n);
        return off; };
    ct.trans = put_thr(&t->trans[0][0], 21, false);          // p < a, p < a + b            (S:1860-1864)
    ct.mix_w = put_thr(t->mix_w, 3, false);                   // tmp_rand < weight           (mm:44, 54)
    for (int ty = 0; ty < 3; ++ty)                            // p > cdf[v]: walk of the inverse-CDF tables
        for (int c = 0; c < 2; ++c) {
            ct.mix_n[ty][c] = t->mix_n[ty][c]; ct.mix_cdf[ty][c] = put_thr(t->mix_cdf[ty][c], t->mix_n[ty][c], true);
            // guide of the walk by the number of leading one bits of the draw (the tail of a run-length CDF is geometric: a constant
            // number of thresholds per halving of 1 - p): g[l] = thresholds at or below the smallest draw with l leading ones — a
            // lower bound of the walk's result for every draw of that class, so the walk starts there instead of at 0
            uint8_t g2[40] = {0};
            const uint64_t *G = blob.data() + ct.mix_cdf[ty][c];
            const uint32_t nn = t->mix_n[ty][c];
            for (uint32_t l = 0; l <= 32; ++l) {
                const uint64_t lo_u = l == 0 ? 0ull : (0xffffffffull << (32 - l)) & 0xffffffffull;
                uint32_t v = 0;
                while (v + 1 < nn && lo_u >= G[v]) ++v;
                g2[l] = (uint8_t)(v > 255u ? 255u : v);
            }
            ct.mix_g2[ty][c] = put_raw(g2, 40);
        }
    ct.fm_n = t->fm_nseg; ct.fm_vlo0 = t->fm_vlo0;
    ct.fm_g = put_thr(t->fm_hi, t->fm_nseg, true);           // p > hi[s]: the segment search of the ECDF look-ups (ecdf_lookup_u)
    { auto g = guide(t->fm_hi, t->fm_nseg); ct.fm_guide = put_raw(g.data(), 512); }
    ct.mm_nbins = t->mm_nbins;
    std::vector<int32_t> bins(2 * (size_t)t->mm_nbins);
    for (uint32_t b = 0; b < t->mm_nbins; ++b) {
        auto clamp = [](int64_t v) { return (int32_t)(v > 0x7fffffff ? 0x7fffffff : v < -0x7fffffff ? -0x7fffffff : v); };
        bins[2 * b] = clamp(t->mm_bin_lo[b]); bins[2 * b + 1] = clamp(t->mm_bin_hi[b]);
    }
    ct.mm_bin = put_raw(bins.data(), bins.size() * 4);
    {   // direct bin of a previous match length < 256 (first bin with lo <= v < hi, else the last bin, S:1891-1893)
        std::vector<uint8_t> lut(256);
        for (int v = 0; v < 256; ++v) {
            uint32_t b = 0;
            for (; b < t->mm_nbins; ++b) if (bins[2 * b] <= v && v < bins[2 * b + 1]) break;
            if (b >= t->mm_nbins) b = t->mm_nbins - 1;
            lut[v] = (uint8_t)b;
        }
        ct.mm_bin_lut = put_raw(lut.data(), 256);
    }
    ct.mm_seg_off = put_raw(t->mm_seg_off, ((size_t)t->mm_nbins + 1) * 4);
    ct.mm_g = put_thr(t->mm_hi, nseg, true); ct.mm_vlo0 = put_d(t->mm_vlo0, t->mm_nbins);
    std::vector<uint16_t> gall;
    for (uint32_t b = 0; b < t->mm_nbins; ++b) {
        auto g = guide(t->mm_hi + t->mm_seg_off[b], t->mm_seg_off[b + 1] - t->mm_seg_off[b]);
        gall.insert(gall.end(), g.begin(), g.end());
    }
    ct.mm_guide = put_raw(gall.data(), gall.size() * 2);
    // value edges: whole numbers in every model read_analysis.py writes (its bins are "i-(i+1)") -> 32-bit copies for the LDS image;
    // the fp64 originals follow behind the part that is copied to LDS (the cooperative chain and a model with fractional edges read those)
    whole = true;
    // The steps of the interpolation floor((p - plo) / (hs - plo) * (vs - vlo) + vlo) inside a segment, found with the arithmetic of
    // the fp64 formula (this file is compiled with -ffp-contract=off, like the device code and the oracle) — ecdf_lookup_u:
    // bit 31 of a value edge: one unit wide and every draw gives vlo; else, up to 15 units wide: thresholds in `sub`, their number
    // and position in the upper bits of the segment's G word
    std::vector<uint64_t> sub;
    auto put_u = [&](uint32_t g_off, const double *hi, const double *src, size_t n, double vlo0, std::vector<uint32_t> &v) {
        for (size_t i = 0; i < n; ++i) {
            if (!(src[i] >= 0 && src[i] < 2147483648.0 && src[i] == floor(src[i]))) whole = false;
            uint32_t e = (uint32_t)(src[i] < 0 ? 0 : src[i] >= 2147483647.0 ? 2147483647.0 : src[i]);
            const double hs = hi[i], plo = i ? hi[i - 1] : 0.0, vs = src[i], vlo = i ? src[i - 1] : vlo0;
            const double w = vs - vlo;
            if (w >= 1.0 && w <= 15.0 && w == floor(w) && hs > plo && vlo == floor(vlo) && vlo >= 0) {
                auto f = [&](uint64_t u) { const double pp = u32_to_p((uint32_t)u); return floor((pp - plo) / (hs - plo) * (vs - vlo) + vlo); };
                const uint64_t u0 = i ? ns_thr_gt(plo) : 0ull, u1 = ns_thr_gt(hs);     // the draws of the segment: [u0, u1)
                uint64_t thr[15];
                for (uint32_t k = 1; k <= (uint32_t)w; ++k) {                          // smallest draw of the segment that gives >= vlo + k
                    uint64_t lo = u0, hi2 = u1;                                         // (f is non-decreasing in the draw)
                    while (lo < hi2) { const uint64_t mid = lo + ((hi2 - lo) >> 1); if (f(mid) >= vlo + (double)k) hi2 = mid; else lo = mid + 1; }
                    thr[k - 1] = lo >= u1 ? (1ull << 32) : lo;
                }
                if (w == 1.0 && thr[0] == (1ull << 32)) e |= 0x80000000u;
                else {
                    blob[g_off + i] |= (uint64_t)(uint32_t)w << 36 | (uint64_t)sub.size() << 40;
                    sub.insert(sub.end(), thr, thr + (uint32_t)w);
                }
            }
            v.push_back(e);
        } };
    { std::vector<uint32_t> v; put_u(ct.fm_g, t->fm_hi, t->fm_vhi, t->fm_nseg, t->fm_vlo0, v); ct.fm_vhi_u = put_raw(v.data(), v.size() * 4); }
    {
        std::vector<uint32_t> v;                               // per column: its first segment starts at the column's vlo0
        for (uint32_t b = 0; b < t->mm_nbins; ++b) {
            const uint32_t o = t->mm_seg_off[b];
            put_u(ct.mm_g + o, t->mm_hi + o, t->mm_vhi + o, t->mm_seg_off[b + 1] - o, t->mm_vlo0[b], v);
        }
        ct.mm_vhi_u = put_raw(v.data(), v.size() * 4);
    }
    if (sub.size() >= (1u << 24)) whole = false;
    sub.push_back(0);
    ct.sub = put_raw(sub.data(), sub.size() * 8);
    ct.n_words_lds = (uint32_t)blob.size();
    ct.fm_hi = put_d(t->fm_hi, t->fm_nseg); ct.mm_hi = put_d(t->mm_hi, nseg);      // fp64 tables: global memory (wide segments, cooperative chain)
    ct.fm_vhi = put_d(t->fm_vhi, t->fm_nseg); ct.mm_vhi = put_d(t->mm_vhi, nseg);
    ct.n_words = (uint32_t)blob.size();
#ifdef NS_CHAIN_TABS2
    ct.mix_rec = ct.fm_gv = ct.mm_gv = ct.pm_lut = ct.sub2 = 0;
    if (layout) ns_pack_layout2(ct, blob, whole, layout, t->fm_nseg, nseg, segs);
#else
    (void)layout;
#endif
}

#ifdef NS_CHAIN_TABS2
// Layouts 1 and 3 (experiments, NS_CHAIN_VAR bits 8 and 32): the image of layout 0 with tables added to its LDS part and, for layout 3,
// the threshold / value-edge / step tables of ecdf_lookup_u moved behind it (the one-word segments replace them in LDS).
//   mix_rec   6 records {table offset | thresholds << 32, guide offset}, index 2 * type + component
//   fm_gv, mm_gv   per ECDF segment ONE word: bits 0..32 the threshold ns_thr_gt(hi[s]); bit 33: one unit wide and every draw gives vlo —
//             bits 35.. = vhi; bit 34: 2..15 units wide (or one unit with a step inside) — bits 35.. = index into sub2 of {vhi | steps << 32}
//             followed by the step thresholds; neither bit: the fp64 formula on the global tables
//   pm_lut    previous match v < 256 -> {first segment of its column | segments << 32 | bin << 56}
static inline void ns_pack_layout2(ChainTab &ct, std::vector<uint64_t> &blob, bool &whole, uint32_t layout, uint32_t fm_n, uint32_t nseg,
                                   const std::vector<std::pair<uint32_t, uint32_t>> &segs) {
    std::vector<uint64_t> fm_gv, mm_gv, pm, sub2;
    if (layout & 2u) {
        auto conv = [&](uint32_t g_off, uint32_t v_off, uint32_t n, std::vector<uint64_t> &out) {
            const uint32_t *vu = reinterpret_cast<const uint32_t *>(blob.data() + v_off);
            for (uint32_t i = 0; i < n; ++i) {
                const uint64_t g = blob[g_off + i], thr = g & 0x1ffffffffull;
                const uint32_t nt = (uint32_t)(g >> 36) & 15u, v = vu[i];
                if ((v & 0x7fffffffu) >= (1u << 29)) whole = false;
                uint64_t word = thr;
                if (v & 0x80000000u) word |= 1ull << 33 | (uint64_t)(v & 0x7fffffffu) << 35;
                else if (nt) {
                    word |= 1ull << 34 | (uint64_t)sub2.size() << 35;
                    sub2.push_back((uint64_t)v | (uint64_t)nt << 32);
                    for (uint32_t k = 0; k < nt; ++k) sub2.push_back(blob[ct.sub + (g >> 40) + k]);
                }
                out.push_back(word);
            } };
        conv(ct.fm_g, ct.fm_vhi_u, fm_n, fm_gv);
        conv(ct.mm_g, ct.mm_vhi_u, nseg, mm_gv);
        if (sub2.size() >= (1u << 28)) whole = false;
        sub2.push_back(0);
        const uint8_t *lut = reinterpret_cast<const uint8_t *>(blob.data() + ct.mm_bin_lut);
        const uint32_t *so = reinterpret_cast<const uint32_t *>(blob.data() + ct.mm_seg_off);
        for (uint32_t v = 0; v < 256; ++v) {
            const uint32_t b = lut[v], o = so[b], nc = so[b + 1] - o;
            if (nc >= (1u << 24)) whole = false;
            pm.push_back((uint64_t)o | (uint64_t)(nc & 0xffffffu) << 32 | (uint64_t)b << 56);
        }
    }
    const uint32_t n_lds1 = ct.n_words_lds;
    auto moved = [&](uint32_t off) { return (layout & 2u) && (off == ct.fm_g || off == ct.mm_g || off == ct.fm_vhi_u || off == ct.mm_vhi_u || off == ct.sub); };
    std::vector<uint64_t> nb;
    std::vector<std::pair<uint32_t, uint32_t>> remap;          // old first word -> new first word
    auto take = [&](const std::pair<uint32_t, uint32_t> &sg) {
        remap.emplace_back(sg.first, (uint32_t)nb.size());
        nb.insert(nb.end(), blob.begin() + sg.first, blob.begin() + sg.first + sg.second); };
    auto add = [&](const std::vector<uint64_t> &v) { const uint32_t off = (uint32_t)nb.size(); nb.insert(nb.end(), v.begin(), v.end()); return off; };
    for (const auto &sg : segs) if (sg.first < n_lds1 && !moved(sg.first)) take(sg);
    uint32_t rec_at = 0;
    if (layout & 1u) rec_at = add(std::vector<uint64_t>(12, 0));
    if (layout & 2u) { ct.fm_gv = add(fm_gv); ct.mm_gv = add(mm_gv); ct.pm_lut = add(pm); ct.sub2 = add(sub2); }
    const uint32_t n_lds = (uint32_t)nb.size();
    for (const auto &sg : segs) if (sg.first < n_lds1 && moved(sg.first)) take(sg);
    for (const auto &sg : segs) if (sg.first >= n_lds1) take(sg);
    auto mv = [&](uint32_t &f) { for (const auto &r : remap) if (r.first == f) { f = r.second; return; } whole = false; };
    mv(ct.trans); mv(ct.mix_w);
    for (int ty = 0; ty < 3; ++ty) for (int c = 0; c < 2; ++c) { mv(ct.mix_cdf[ty][c]); mv(ct.mix_g2[ty][c]); }
    mv(ct.fm_g); mv(ct.fm_guide); mv(ct.mm_bin); mv(ct.mm_bin_lut); mv(ct.mm_seg_off); mv(ct.mm_g); mv(ct.mm_vlo0); mv(ct.mm_guide);
    mv(ct.fm_vhi_u); mv(ct.mm_vhi_u); mv(ct.sub); mv(ct.fm_hi); mv(ct.mm_hi); mv(ct.fm_vhi); mv(ct.mm_vhi);
    if (layout & 1u) {
        ct.mix_rec = rec_at;
        for (uint32_t ty = 0; ty < 3; ++ty) for (uint32_t c = 0; c < 2; ++c) {
            nb[rec_at + 2 * (2 * ty + c)] = (uint64_t)ct.mix_cdf[ty][c] | (uint64_t)ct.mix_n[ty][c] << 32;
            nb[rec_at + 2 * (2 * ty + c) + 1] = ct.mix_g2[ty][c];
        }
    }
    blob.swap(nb);
    ct.n_words_lds = n_lds; ct.n_words = (uint32_t)blob.size();
}
#endif
