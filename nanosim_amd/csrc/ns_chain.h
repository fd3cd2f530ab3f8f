// ns_chain.h — the error-event Markov chains of error_list (S:1833-1916) and unaligned_error_list
// (S:1784-1830) on tables packed into ONE blob of 8-byte words (ns_pack.h) that k_chain copies into LDS.
//   chain_error_list            thread per read on the LDS image: integer thresholds, one word per ECDF segment, the reads of an
//                               iteration issued round by round in one basic block
//   chain_error_list_g          thread per read on the fp64 tables in global memory (models whose value edges are not whole numbers or
//                               whose image does not fit LDS): the reference's arithmetic as it is
//   chain_unaligned_error_list  thread per read, unaligned reads and gaps
//   coop_error_list, coop_unaligned_error_list   one read per wavefront (the longest aligned reads of a batch, gaps; lane = loop iteration)
//   coopk_unaligned_error_list  one unaligned read per wavefront, K loop iterations per lane (all unaligned reads)
// All produce the events of the oracle's error_list / unaligned_error_list bit for bit (tests/test_chain_host.py compiles this source
// for the host; the -m gpu parity tests compare whole batches).
#pragma once
#include "ns_device.h"

struct Tabs {
    const uint64_t *w;
    NS_DEV const double *d(uint32_t off) const { return reinterpret_cast<const double *>(w + off); }
    NS_DEV const uint64_t *q(uint32_t off) const { return w + off; }
    NS_DEV const uint16_t *h(uint32_t off) const { return reinterpret_cast<const uint16_t *>(w + off); }
    NS_DEV const uint32_t *u(uint32_t off) const { return reinterpret_cast<const uint32_t *>(w + off); }
    NS_DEV const int32_t *i(uint32_t off) const { return reinterpret_cast<const int32_t *>(w + off); }
};

// first s with p <= hi[s] (guide[u>>24] is a lower bound of s), then the interpolation of S:1847 / S:1897.
// hi[s] and hi[s+1] are fetched together: the look-up sits on the chain's critical path.
template <class V>       // V = double, or uint32_t for the integer copy of the value edges in LDS (same values, converted on the fly)
NS_DEV int32_t ecdf_lookup_g(const double *__restrict__ hi, const V *__restrict__ vhi, uint32_t n,
                                                 double vlo0, const uint16_t *__restrict__ guide, uint32_t u) {
    double p = u32_to_p(u);
    uint32_t s = guide[u >> 24];
    if (s < n) {
        const double h0 = hi[s], h1 = hi[min(s + 1, n - 1)];
        if (p > h0) {
            ++s;
            if (s < n && p > h1) { ++s; while (s < n && p > hi[s]) ++s; }
        }
    }
    if (s >= n) { s = n - 1; p = hi[s]; }
    const uint32_t sm = s ? s - 1 : 0;
    const double hs = hi[s], hp = hi[sm], vs = (double)vhi[s], vp = (double)vhi[sm];
    const double plo = s ? hp : 0.0;
    const double vlo = s ? vp : vlo0;
    return (int32_t)floor((p - plo) / (hs - plo) * (vs - vlo) + vlo);
}

// mixture run length (mm:41-63) on integer thresholds: component by u_mix < T(weight), value = 1 + #{j : p > cdf[j]} by walking
// G[j] = ns_thr_gt(cdf[j]) — no fp64 on the way.  Offset, length and guide of the table of (type, component) come from its record in the
// blob: ONE read (as fields of ChainTab indexed by a per-thread type they were three vector loads from the kernel-argument segment on
// every event's critical path — found in the ISA of round 4).
NS_DEV void mix_record(const Tabs &T, const ChainTab &c, uint32_t type, uint32_t comp, uint32_t &go, uint32_t &n, uint32_t &h) {
    const uint64_t *r = T.w + c.mix_rec + 2u * (2u * type + comp);
    const uint64_t r0 = r[0], r1 = r[1];
    go = (uint32_t)r0; n = (uint32_t)(r0 >> 32); h = (uint32_t)r1;
}
NS_DEV int32_t run_length_t(const Tabs &T, const ChainTab &c, int type, uint32_t u_mix, uint32_t u_len) {
    const int comp = ((uint64_t)u_mix < T.q(c.mix_w)[type]) ? 0 : 1;         // tmp_rand < weight, mm:44,54
    uint32_t go, n, h;
    mix_record(T, c, (uint32_t)type, (uint32_t)comp, go, n, h);
    const uint64_t u = u_len;
    // == v = 0; while (v + 1 < n && p > cdf[v]) ++v — started at the guide's lower bound for draws with this many leading one bits
    // (every threshold below it is <= the smallest such draw): zero to two steps instead of (run length - 1) dependent reads
    uint32_t v = reinterpret_cast<const uint8_t *>(T.w)[8u * h + ns_clz32(~u_len)];        // (__clz(0) == 32: the draw 0xffffffff)
    while (v + 1 < n && u >= T.w[go + v]) ++v;
    return (int32_t)v + 1;
}

struct EvSink32 {
    ns_event *ev;
    uint32_t cap, n;
    int32_t shift;
    uint32_t last_ins_len;
    bool overflow;
    bool range;          // an event does not fit the 8-byte record: run longer than NS_EV_LEN_MAX, or |shift| >= NS_EV_SHIFT_BIAS
    // thread-per-read chain: LDS column of this thread, four slots `stride` (= threads of the workgroup) apart (nullptr: events are stored one by one).
    // A thread's scattered 8-byte stores cost a 32-byte memory write each (WRITE_SIZE 10.3 KB per read for 2.1 KB of events); staged,
    // four events leave as one aligned 32-byte group (ev and cap are multiples of four events then)
    uint2 *stg;
    uint32_t stride;
};
#ifndef NS_CHAIN_BLOCK
#define NS_CHAIN_BLOCK 256
#endif
NS_DEV void ev_flush4(EvSink32 &s, uint32_t first) {       // staged slots 0..3 -> events first .. first + 3
    const uint2 e0 = s.stg[0], e1 = s.stg[s.stride], e2 = s.stg[2u * s.stride], e3 = s.stg[3u * s.stride];
    uint4 *dst = reinterpret_cast<uint4 *>(s.ev + first);
    // (plain stores: as nontemporal stores these scattered 32-byte groups took the thread-per-read chain from 3.2 to 8.5 ms,
    // profiles/r05/ab_nt_more.log — nontemporal pays for the wave-wide contiguous streams of the record and error-profile images)
    dst[0] = make_uint4(e0.x, e0.y, e1.x, e1.y); dst[1] = make_uint4(e2.x, e2.y, e3.x, e3.y);
}
// the events still staged when a piece is complete (slots behind the last event carry stale values: inside the capacity, never read)
NS_DEV void ev_flush_tail(EvSink32 &s) {
    if (s.stg && (s.n & 3u) && s.n < s.cap) ev_flush4(s, s.n & ~3u);
}
// does the cumulative shift fit the 18-bit field of ns_event.info?
NS_DEV bool ev_shift_fits(int32_t shift) { return (uint32_t)(shift + NS_EV_SHIFT_BIAS) < 2u * (uint32_t)NS_EV_SHIFT_BIAS; }
NS_DEV void ev_push32(EvSink32 &s, int32_t pos, uint32_t type, int32_t len) {
    uint32_t l = len > (int32_t)NS_EV_LEN_MAX ? NS_EV_LEN_MAX : (uint32_t)len;
    if (len > (int32_t)NS_EV_LEN_MAX || !ev_shift_fits(s.shift)) s.range = true;
    if (s.n < s.cap) {
        ns_event e; e.pos = (uint32_t)pos; e.info = ns_ev_pack(l, type, s.shift);
        if (s.stg) {
            s.stg[(s.n & 3u) * s.stride] = make_uint2(e.pos, e.info);
            if ((s.n & 3u) == 3u) ev_flush4(s, s.n - 3u);
        } else s.ev[s.n] = e;
    } else s.overflow = true;
    if (type == NS_INS) { s.shift += (int32_t)l; s.last_ins_len = l; } else if (type == NS_DEL) s.shift -= (int32_t)l;
    s.n++;
}

struct EList32 { int32_t l_new, middle_ref; };

// error_list, S:1833-1916, on the fp64 tables (T = the whole blob in global memory): fp64 compares + the fp64 interpolation of the
// reference, every search started from a 256-entry guide index instead of a full binary search
NS_DEV EList32 chain_error_list_g(const Tabs &T, const ChainTab &c, int32_t m_ref, const ns_key &key, uint32_t seg, uint32_t attempt, EvSink32 &s) {
    int32_t l_new = m_ref, pos = 0, middle_ref = m_ref;
    int state = NS_ST_START;
    u32x4 w = ns_draw(key, ST_EVENT, seg, attempt, 0, 0);
    int32_t prev_match = ecdf_lookup_g(T.d(c.fm_hi), T.d(c.fm_vhi), c.fm_n, c.fm_vlo0, T.h(c.fm_guide), w.x);          // S:1843-1850
    if (prev_match < 2) prev_match = 2;
    pos += prev_match;
    uint32_t it = 1;
    int32_t last_ins_pos = -1;
    const uint64_t *trans = T.q(c.trans);
    const int32_t *bins = T.i(c.mm_bin);
    const uint32_t *seg_off = T.u(c.mm_seg_off);
    const uint8_t *bin_lut = reinterpret_cast<const uint8_t *>(T.w + c.mm_bin_lut);
    u32x4 w_next = ns_draw(key, ST_EVENT, seg, attempt, 1, 0);
    while (pos < middle_ref) {                                                                     // S:1858
        w = w_next;
        w_next = ns_draw(key, ST_EVENT, seg, attempt, it + 1, 0);    // next iteration's draws do not depend on the chain state
        const int error = trans_pick_u(trans + 3 * state, w.x);                                   // S:1860-1864
        int32_t step = run_length_t(T, c, error, w.y, w.z);                                        // S:1866-1873
        if (error == NS_INS) l_new += step; else if (error == NS_DEL) l_new -= step;
        if (error != NS_INS) {                                                                     // S:1875-1880
            ev_push32(s, pos, (uint32_t)error, step);
            pos += step;
            if (pos >= middle_ref) { l_new += pos - middle_ref; middle_ref = pos; }
        } else {                                                                                   // S:1881-1882
            if (last_ins_pos == pos && s.n > 0) { s.n--; s.shift -= (int32_t)s.last_ins_len; }    // dict key collision
            ev_push32(s, pos, NS_INS, step);
            last_ins_pos = pos;
        }
        state = NS_ST_MIS + error;                                                                 // S:1884
        uint32_t b;                                                                                // S:1891-1893
        if ((uint32_t)prev_match < 256u) b = bin_lut[prev_match];
        else {
            for (b = 0; b < c.mm_nbins; ++b)
                if (bins[2 * b] <= prev_match && prev_match < bins[2 * b + 1]) break;
            if (b >= c.mm_nbins) b = c.mm_nbins - 1;
        }
        const uint32_t o = seg_off[b], ncol = seg_off[b + 1] - o;
        step = ecdf_lookup_g(T.d(c.mm_hi) + o, T.d(c.mm_vhi) + o, ncol, T.d(c.mm_vlo0)[b], T.h(c.mm_guide) + 256 * b, w.w);
        if (prev_match == 0 && step == 0) step = 1;                                                // S:1900-1901
        prev_match = step;
        if (pos + prev_match > middle_ref) { l_new += pos + prev_match - middle_ref; middle_ref = pos + prev_match; }
        pos += prev_match;
        if (prev_match == 0) state += 3;                                                           // S:1913-1914
        else last_ins_pos = -1;
        ++it;
    }
    return EList32{l_new, middle_ref};
}

// ---- error_list on the LDS image ---------------------------------------------------------------------------------------------------
// What the ISA and the counters of the round-3/4 chain showed (profiles/r04): an event cost ~270 vector + ~180 scalar instructions and
// eleven DEPENDENT look-ups — transition row, three loads from the kernel-argument segment, guide byte, one or two thresholds; bin byte,
// segment range, guide, one or two thresholds, value edge — with four wavefronts per SIMD to hide them.  Here
//   * a segment is ONE word (threshold, class, value: ns_pack.h), so the look-up ends with the threshold it stops at;
//   * the column of the previous match comes from one word (pm_lut) instead of bin byte -> segment range;
//   * the next match length depends on the previous match and the draw only: it is looked up FIRST, next to the run length, and both
//     run ahead of the event store (whose LDS staging writes the compiler cannot move table reads across) — two chains of three reads
//     side by side behind the transition row instead of eleven in a row;
//   * one event-store site and the run-length record.
// Round 5, same box (profiles/r05/ab_chain.log): aligned k_chain 3.28 -> 2.94 ms alone, the thread-per-read unaligned chain next to the
// aligned call 8.8 -> 3.8 ms, whole step 11.3 -> 9.9 ms.
//
// ev_push32 with the bookkeeping behind the store as selects (no exec-mask regions for the type) and the two flags as ARITHMETIC: the
// longest run and the extremes of the shift at store time are tracked (three vector instructions) and turned into `range` / `overflow`
// once per piece (ev_track_close) — kept as booleans they are lane masks the compiler updates with three scalar instructions per flag
// and store site.  overflow == "n ended above the capacity": n never ends an iteration below where it started (the dict-key collision
// pops one event and pushes one).
struct EvTrack { int32_t max_len, smin, smax; };
NS_DEV EvTrack ev_track_open(const EvSink32 &s) { return EvTrack{0, s.shift, s.shift}; }
NS_DEV void ev_track_close(EvSink32 &s, const EvTrack &t) {
    if (t.max_len > (int32_t)NS_EV_LEN_MAX || !ev_shift_fits(t.smin) || !ev_shift_fits(t.smax)) s.range = true;
    if (s.n > s.cap) s.overflow = true;
}
NS_DEV void ev_push32s(EvSink32 &s, EvTrack &t, int32_t pos, uint32_t type, int32_t len) {
    const uint32_t l = len > (int32_t)NS_EV_LEN_MAX ? NS_EV_LEN_MAX : (uint32_t)len;
    t.max_len = len > t.max_len ? len : t.max_len;
    t.smin = s.shift < t.smin ? s.shift : t.smin; t.smax = s.shift > t.smax ? s.shift : t.smax;
    if (s.n < s.cap) {
        ns_event e; e.pos = (uint32_t)pos; e.info = ns_ev_pack(l, type, s.shift);
        if (s.stg) {
            s.stg[(s.n & 3u) * s.stride] = make_uint2(e.pos, e.info);
            if ((s.n & 3u) == 3u) ev_flush4(s, s.n - 3u);
        } else s.ev[s.n] = e;
    }
    s.shift += type == NS_INS ? (int32_t)l : type == NS_DEL ? -(int32_t)l : 0;
    s.last_ins_len = type == NS_INS ? l : s.last_ins_len;
    s.n++;
}

// run length (S:1866-1873 / S:1796-1806) by the record of (type, component): record -> guide byte -> the first two thresholds read
// TOGETHER (the word behind a table is read, never used) -> arithmetic compares; a walk of more than two steps is the rare tail
NS_DEV int32_t run_length_r(const Tabs &T, const ChainTab &c, uint32_t type, uint64_t weight_thr, uint32_t u_mix, uint32_t u_len) {
    const uint32_t comp = (uint64_t)u_mix >= weight_thr ? 1u : 0u;                                 // tmp_rand < weight, mm:44,54
    const uint64_t *r = T.w + c.mix_rec + 2u * (2u * type + comp);
    const uint64_t r0 = r[0], r1 = r[1];
    const uint32_t go = (uint32_t)r0, rn = (uint32_t)(r0 >> 32);
    const uint64_t uz = u_len;
    uint32_t rv = reinterpret_cast<const uint8_t *>(T.w)[8u * (uint32_t)r1 + ns_clz32(~u_len)];
    const uint64_t ra = T.w[go + rv], rb = T.w[go + rv + 1u];
    const uint32_t c1 = (rv + 1u < rn ? 1u : 0u) & (uz >= ra ? 1u : 0u), c2 = c1 & (rv + 2u < rn ? 1u : 0u) & (uz >= rb ? 1u : 0u);
    rv += c1 + c2;
    if (c2) while (rv + 1u < rn && uz >= T.w[go + rv]) ++rv;
    return (int32_t)rv + 1;
}

#define NS_G_THR(g) ((g) & 0x1ffffffffull)
#define NS_GV_UNIT (1ull << 33)
#define NS_GV_NARROW (1ull << 34)
// the segment of draw u in a column of n one-word segments and the value it gives.  Straight-line for the common case — the guide's
// segment or the next, one unit wide: the two words are read TOGETHER and unconditionally (GV[s + 1] behind the column's end is the next
// table's first word: read, never used), the compares are arithmetic, nothing the compiler could sink a load into
NS_DEV int32_t ecdf_lookup_gv(const uint64_t *__restrict__ GV, uint32_t n, const uint16_t *__restrict__ guide, uint32_t u,
                              const uint64_t *__restrict__ sub2, const double *__restrict__ hi_g, const double *__restrict__ vhi_g, double vlo0) {
    const uint32_t s0 = guide[u >> 24];
    const uint64_t uu = u;
    const uint32_t sc = min(s0, n - 1u);
    const uint64_t g0 = GV[sc], g1 = GV[sc + 1u];
    const uint32_t k0 = (s0 < n) & (uu >= NS_G_THR(g0)), k1 = k0 & (s0 + 1u < n) & (uu >= NS_G_THR(g1));
    uint64_t g = k0 ? g1 : g0;
    uint32_t s = s0 + k0 + k1;
    if (k1) {                                     // three or more segments inside one guide cell: bisection between the guide's bounds — the
        const uint32_t cell = u >> 24;            // last cells of a trained model's column hold hundreds of segments (its tail rows), and in the
        uint32_t hi2 = cell < 255u ? min((uint32_t)guide[cell + 1u], n) : n;      // full column every step of a walk is a global-memory read
        while (s < hi2) { const uint32_t mid = (s + hi2) >> 1; if (uu >= NS_G_THR(GV[mid])) s = mid + 1u; else hi2 = mid; }
        if (s < n) g = GV[s];
    }
    if (s < n && (g & NS_GV_UNIT)) return (int32_t)(uint32_t)(g >> 35) - 1;
    if (s < n && (g & NS_GV_NARROW)) {
        const uint64_t *t = sub2 + (g >> 35);
        const uint64_t h = t[0];
        const uint32_t nt = (uint32_t)(h >> 32);
        int32_t r = (int32_t)(uint32_t)h - (int32_t)nt;
        for (uint32_t k = 1; k <= nt; ++k) r += uu >= t[k] ? 1 : 0;
        return r;
    }
    double p = u32_to_p(u);                                  // a wide segment, or a draw above the last edge (clamped to it: frac = 1 -> vhi)
    if (s >= n) { s = n - 1; p = hi_g[s]; }
    const uint32_t sm = s ? s - 1 : 0;
    const double hs = hi_g[s], plo = s ? hi_g[sm] : 0.0, vs = vhi_g[s], vlo = s ? vhi_g[sm] : vlo0;
    return (int32_t)floor((p - plo) / (hs - plo) * (vs - vlo) + vlo);
}

// the same walk on the PREFIX of a column (the LDS image, ns_pack.h): answers for a unit-wide or narrow segment inside the prefix; false: the
// draw lies behind the prefix, or in a wide segment (fp64 formula) — the full column decides
NS_DEV bool ecdf_lookup_pre(const uint64_t *__restrict__ GV, uint32_t n, const uint16_t *__restrict__ guide, uint32_t u,
                            const uint64_t *__restrict__ sub2, int32_t &out) {
    const uint64_t uu = u;
    const uint32_t cell = u >> 24;
    uint32_t s = guide[cell], hi2 = cell < 255u ? min((uint32_t)guide[cell + 1u], n) : n;       // first segment with u < threshold: bisection
    while (s < hi2) { const uint32_t mid = (s + hi2) >> 1; if (uu >= NS_G_THR(GV[mid])) s = mid + 1u; else hi2 = mid; }   // between the guide's bounds
    if (s >= n) return false;
    const uint64_t g = GV[s];
    if (g & NS_GV_UNIT) { out = (int32_t)(uint32_t)(g >> 35) - 1; return true; }
    if (g & NS_GV_NARROW) {
        const uint64_t *t = sub2 + (g >> 35);
        const uint64_t h = t[0];
        const uint32_t nt = (uint32_t)(h >> 32);
        int32_t r = (int32_t)(uint32_t)h - (int32_t)nt;
        for (uint32_t k = 1; k <= nt; ++k) r += uu >= t[k] ? 1 : 0;
        out = r;
        return true;
    }
    return false;
}
// the generic look-up of the next match length (any previous match, any segment class, any draw): the fall-back of chain_error_list's fast
// path.  The prefix column in the LDS image first (a walk of more than two segments, a narrow segment: a few percent of the events); the
// FULL column in global memory for what the prefix cannot answer (a draw behind it: one in 2^tail_bits; a wide segment; a previous match >= 256)
NS_DEV int32_t next_match_gv(const Tabs &T, const Tabs &TG, const ChainTab &c, int32_t prev_match, uint32_t u) {
    uint32_t b, o, ncol;
    if ((uint32_t)prev_match < 256u) {
        const uint64_t pp = T.q(c.pm_lut)[prev_match];
        int32_t r;
        if (ecdf_lookup_pre(T.q(c.mm_gv) + (uint32_t)pp, (uint32_t)(pp >> 32) & 0xffffffu, T.h(c.mm_guide) + 256u * (uint32_t)(pp >> 56), u, T.q(c.sub2), r)) return r;
        const uint64_t pe = TG.q(c.pm_full)[prev_match];
        b = (uint32_t)(pe >> 56); o = (uint32_t)pe; ncol = (uint32_t)(pe >> 32) & 0xffffffu;
    } else {                                                                                       // S:1891-1893
        const int32_t *bins = T.i(c.mm_bin);
        const uint32_t *seg_off = T.u(c.mm_seg_off);
        for (b = 0; b < c.mm_nbins; ++b)
            if (bins[2 * b] <= prev_match && prev_match < bins[2 * b + 1]) break;
        if (b >= c.mm_nbins) b = c.mm_nbins - 1;
        // (a trained model's matches run to thousands of bases: one event in a few thousand follows a match >= 256 — one iteration in fifty
        // of a wavefront; its prefix column answers from LDS like any other)
        const uint64_t pp = T.q(c.pm_bin)[b];
        int32_t r;
        if (ecdf_lookup_pre(T.q(c.mm_gv) + (uint32_t)pp, (uint32_t)(pp >> 32) & 0xffffffu, T.h(c.mm_guide) + 256u * b, u, T.q(c.sub2), r)) return r;
        o = seg_off[b]; ncol = seg_off[b + 1] - o;
    }
    return ecdf_lookup_gv(TG.q(c.mm_gv_full) + o, ncol, T.h(c.mm_guide) + 256u * b, u, TG.q(c.sub2_full), TG.d(c.mm_hi) + o, TG.d(c.mm_vhi) + o, T.d(c.mm_vlo0)[b]);
}

// T: the LDS image (the first n_words_lds words of the blob); TG: the whole blob in global memory (fp64 tables of the wide segments)
NS_DEV EList32 chain_error_list(const Tabs &T, const Tabs &TG, const ChainTab &c, int32_t m_ref, const ns_key &key,
                                uint32_t seg, uint32_t attempt, EvSink32 &s) {
    int32_t l_new = m_ref, pos = 0, middle_ref = m_ref;
    uint32_t state = NS_ST_START;
    u32x4 w = ns_draw(key, ST_EVENT, seg, attempt, 0, 0);
    int32_t prev_match = ecdf_lookup_gv(TG.q(c.fm_gv), c.fm_n, T.h(c.fm_guide), w.x, TG.q(c.sub2_full), TG.d(c.fm_hi), TG.d(c.fm_vhi), c.fm_vlo0);   // S:1843-1850 (once per piece: global memory)
    if (prev_match < 2) prev_match = 2;
    pos += prev_match;
    uint32_t it = 1;
    int32_t last_ins_pos = -1;
    const uint64_t *trans = T.q(c.trans);
    const uint64_t *pm = T.q(c.pm_lut);
    const uint64_t *rec = T.q(c.mix_rec);
    const uint64_t *gv = T.q(c.mm_gv);
    const uint16_t *guide = T.h(c.mm_guide);
    const uint8_t *bytes = reinterpret_cast<const uint8_t *>(T.w);
    const uint64_t mw0 = T.q(c.mix_w)[0], mw1 = T.q(c.mix_w)[1], mw2 = T.q(c.mix_w)[2];
    u32x4 w_next = ns_draw(key, ST_EVENT, seg, attempt, 1, 0);
    EvTrack trk = ev_track_open(s);
    while (pos < middle_ref) {                                                                     // S:1858
        w = w_next;
        w_next = ns_draw(key, ST_EVENT, seg, attempt, it + 1, 0);    // next iteration's draws do not depend on the chain state
        const uint64_t ux = w.x, uy = w.y, uz = w.z, uw = w.w;
        // The iteration asks the tables two independent questions: the run length of this event (transition row -> record of (type,
        // component) -> guide byte -> two thresholds) and the next match length, which depends on the PREVIOUS match and the draw only
        // (column word -> guide -> two segment words).  Their reads are issued side by side, round by round, in one basic block; whatever
        // does not fit the common case — a walk of more than two steps, a segment that is not one unit wide, a draw beyond the column, a
        // previous match >= 256 — is recomputed by the generic functions behind it.
        // ---- round 1: transition row (S:1860-1864); column of the previous match (S:1891-1893)
        const uint64_t t0 = trans[3u * state], t1 = trans[3u * state + 1u];
        const uint32_t small = (uint32_t)prev_match < 256u ? 1u : 0u;
        const uint64_t pe = pm[small ? (uint32_t)prev_match : 0u];
        const uint32_t e0 = ux >= t0 ? 1u : 0u, e1 = ux >= t1 ? 1u : 0u;
        const uint32_t error = e0 + (e0 & e1);                                                     // mis [0, t0), ins [t0, t1), del: the rest (trans_pick_u)
        const uint32_t b = (uint32_t)(pe >> 56), o = (uint32_t)pe, ncol = (uint32_t)(pe >> 32) & 0xffffffu;
        // ---- round 2: record of the run-length table; guide of the column
        const uint64_t mw = error == NS_MIS ? mw0 : error == NS_INS ? mw1 : mw2;
        const uint32_t comp = uy >= mw ? 1u : 0u;                                                  // tmp_rand < weight, mm:44,54
        const uint64_t r0 = rec[2u * (2u * error + comp)], r1 = rec[2u * (2u * error + comp) + 1u];
        const uint32_t s0 = guide[256u * b + (w.w >> 24)];
        // ---- round 3: guide byte of the threshold walk; the guide's segment and the next
        const uint32_t go = (uint32_t)r0, rn = (uint32_t)(r0 >> 32);
        uint32_t rv = bytes[8u * (uint32_t)r1 + ns_clz32(~w.z)];
        const uint32_t sc = min(s0, ncol - 1u);
        const uint64_t g0 = gv[o + sc], g1 = gv[o + sc + 1u];                                      // (a word behind a table is read, never used)
        // ---- round 4: the first two thresholds of the walk
        const uint64_t ra = T.w[go + rv], rb = T.w[go + rv + 1u];
        // ---- answers
        const uint32_t k0 = (s0 < ncol ? 1u : 0u) & (uw >= NS_G_THR(g0) ? 1u : 0u), k1 = k0 & (s0 + 1u < ncol ? 1u : 0u) & (uw >= NS_G_THR(g1) ? 1u : 0u);
        const uint64_t g = k0 ? g1 : g0;
        const uint32_t fast = small & (s0 + k0 < ncol ? 1u : 0u) & (k1 ^ 1u) & ((g & NS_GV_UNIT) ? 1u : 0u);
        int32_t next = (int32_t)(uint32_t)(g >> 35) - 1;
        const uint32_t c1 = (rv + 1u < rn ? 1u : 0u) & (uz >= ra ? 1u : 0u), c2 = c1 & (rv + 2u < rn ? 1u : 0u) & (uz >= rb ? 1u : 0u);
        rv += c1 + c2;
        if (c2) while (rv + 1u < rn && uz >= T.w[go + rv]) ++rv;                                   // (rare: a run longer than the guide's bound + 2)
        if (!fast) next = next_match_gv(T, TG, c, prev_match, w.w);
        const int32_t step = (int32_t)rv + 1;                                                      // S:1866-1873
        if (prev_match == 0 && next == 0) next = 1;                                                // S:1900-1901
        // ---- the event (S:1875-1882), one store site
        const bool ins = error == NS_INS;
        l_new += ins ? step : error == NS_DEL ? -step : 0;
        const bool coll = ins && last_ins_pos == pos && s.n > 0;                                   // dict key collision
        s.n -= coll ? 1u : 0u; s.shift -= coll ? (int32_t)s.last_ins_len : 0;
        ev_push32s(s, trk, pos, error, step);
        last_ins_pos = ins ? pos : last_ins_pos;
        pos += ins ? 0 : step;
        const bool over = !ins && pos >= middle_ref;
        l_new += over ? pos - middle_ref : 0; middle_ref = over ? pos : middle_ref;
        state = NS_ST_MIS + error;                                                                 // S:1884
        prev_match = next;
        if (pos + prev_match > middle_ref) { l_new += pos + prev_match - middle_ref; middle_ref = pos + prev_match; }
        pos += prev_match;
        if (prev_match == 0) state += 3;                                                           // S:1913-1914
        else last_ins_pos = -1;
        ++it;
    }
    ev_track_close(s, trk);
    return EList32{l_new, middle_ref};
}

// unaligned_error_list (S:1784-1830, event rewrite of DESIGN.md section 5.3) in the same style: the seven event-store
// sites a literal restatement has — one per case of (type, pending insertion) — are THREE slots filled by selects: A the
// mismatch / deletion at pos, B the pending insertion behind it at pos + 1, C the rest of a mismatch run behind that insertion; an
// insertion draws no event and does not advance (it waits in pend_ins), without leaving the iteration early.
NS_DEV EList32 chain_unaligned_error_list(const Tabs &T, const ChainTab &c, int32_t m_ref, const ns_key &key,
                                          uint32_t seg, uint32_t attempt, EvSink32 &s) {
    int32_t l_new = m_ref, pos = 0, middle_ref = m_ref;
    int32_t pend_ins = 0;
    if (m_ref <= 0) return EList32{l_new, middle_ref};
    uint32_t it = 0;
    u32x4 w_next = ns_draw(key, ST_UEVENT, seg, attempt, 0, 0);
    const uint64_t mw0 = T.q(c.mix_w)[0], mw1 = T.q(c.mix_w)[1], mw2 = T.q(c.mix_w)[2];
    EvTrack trk = ev_track_open(s);
    while (pos < middle_ref) {
        const u32x4 w = w_next;
        ++it;
        w_next = ns_draw(key, ST_UEVENT, seg, attempt, it, 0);
        const uint64_t ut = w.x;                                                                     // (p < t  <=>  u < ns_thr_lt(t))
        const uint32_t type = (ut < ns_thr_lt(0.4)) ? 3u : (ut < ns_thr_lt(0.7)) ? (uint32_t)NS_MIS : (ut < ns_thr_lt(0.85)) ? (uint32_t)NS_INS : (uint32_t)NS_DEL;   // S:1787
        const uint32_t tt = type == 3u ? (uint32_t)NS_MIS : type;                                    // (a match asks the mismatch table and drops the answer)
        const int32_t run = run_length_r(T, c, tt, tt == NS_MIS ? mw0 : tt == NS_INS ? mw1 : mw2, w.y, w.z);
        const int32_t step = type == 3u ? 1 : run;
        const bool is_ins = type == NS_INS, is_mis = type == NS_MIS, is_del = type == NS_DEL;
        l_new += is_ins ? step : is_del ? -step : 0;                                                 // S:1808-1815
        const int32_t L = is_ins ? 0 : pend_ins;
        pend_ins = is_ins ? pend_ins + step : 0;
        const bool va = is_mis || is_del;
        int32_t dl = step - L; dl = dl < 1 ? 1 : dl;
        const int32_t la = L == 0 ? step : is_mis ? 1 : dl;
        const int32_t lb = is_del ? L - (step - 1) : L;
        const bool vb = L > 0 && lb > 0;
        const bool vc = is_mis && L > 0 && step - 1 > L;
        if (va) ev_push32s(s, trk, pos, type, la);
        if (vb) ev_push32s(s, trk, pos + 1, NS_INS, lb);
        if (vc) ev_push32s(s, trk, pos + 1, NS_MIS, step - 1 - L);
        pos += is_ins ? 0 : step;
        const bool over = pos > middle_ref;                                                          // S:1826-1828
        l_new += over ? pos - middle_ref : 0; middle_ref = over ? pos : middle_ref;
    }
    ev_track_close(s, trk);
    return EList32{l_new, middle_ref};
}

// ---- cooperative unaligned_error_list: one read (or gap) per wavefront, lane = loop iteration -------------------------
// What iteration `it` of S:1797-1829 draws (type, run length) does not depend on the state of the loop, and the state is a running
// sum: pos(it) = sum of the steps of the earlier non-insertion iterations, the loop runs while pos < m_ref.  So 64 iterations are
// evaluated at once: wavefront prefix sums give the positions, the pending insertion lengths (consecutive insertion iterations in
// front of a non-insertion one), the event slots and the cumulative shifts; the events come out exactly as
// chain_unaligned_error_list produces them.  A 20 kb unaligned read is 20 000 dependent iterations for one thread (tens of
// milliseconds: it set the duration of the whole batch) and ~320 blocks of a few hundred instructions here.
__device__ inline EList32 coop_unaligned_error_list(const Tabs &T, const ChainTab &c, int32_t m_ref, const ns_key &key, uint32_t seg,
                                                    uint32_t attempt, EvSink32 &s, uint32_t lane) {
    int32_t l_new = m_ref, middle_ref = m_ref;
    if (m_ref <= 0) return EList32{l_new, middle_ref};
    uint32_t pos0 = 0, pend0 = 0;                                // position / pending insertion length in front of the block
    auto draw = [&](uint32_t it, int &type, uint32_t &step) {    // what iteration `it` does (S:1787, 1799-1818)
        const u32x4 w = ns_draw(key, ST_UEVENT, seg, attempt, it, 0);
        const uint64_t ut = w.x;                                     // (p < t  <=>  u < ns_thr_lt(t))
        type = (ut < ns_thr_lt(0.4)) ? 3 : (ut < ns_thr_lt(0.7)) ? NS_MIS : (ut < ns_thr_lt(0.85)) ? NS_INS : NS_DEL;
        step = 1;
        if (type != 3) step = (uint32_t)run_length_t(T, c, type, w.y, w.z);
    };
    int type_n; uint32_t step_n;
    draw(lane, type_n, step_n);
    for (uint32_t it0 = 0;; it0 += 64) {
        int type = type_n; uint32_t step = step_n;
        draw(it0 + 64 + lane, type_n, step_n);                   // the next block's draws and table reads run under this block's prefix sums
        const uint32_t adv = type == NS_INS ? 0u : step;
        const uint32_t adv_incl = wave_incl_scan(adv);
        const uint32_t pos = pos0 + adv_incl - adv;
        const bool exec = pos < (uint32_t)m_ref;                                                 // the loop is still running (a prefix of the lanes, >= 1)
        const uint32_t n_exec = (uint32_t)__popcll(__ballot(exec));
        if (!exec) { step = 0; type = 3; }
        const uint32_t ins = (exec && type == NS_INS) ? step : 0u, del = (exec && type == NS_DEL) ? step : 0u;
        const uint32_t ins_incl = wave_incl_scan(ins);
        // pending insertion in front of a non-insertion iteration: the insertion steps since the previous non-insertion one
        const bool nonins = exec && type != NS_INS;
        const uint64_t NB = __ballot(nonins);
        const uint64_t below = NB & ((1ull << lane) - 1ull);
        const uint32_t prev_ins = (uint32_t)__shfl((int)ins_incl, below ? 63 - __clzll((long long)below) : (int)lane);
        const uint32_t L = nonins ? (below ? ins_incl - prev_ins : pend0 + ins_incl) : 0u;
        // the events of the iteration, at most three
        uint32_t ty0 = 0, ty1 = 0, ty2 = 0, ln0 = 0, ln1 = 0, ln2 = 0, ps0 = pos, ps1 = pos + 1, ps2 = pos + 1, n_ev = 0;
        if (nonins) {
            if (type == 3) { if (L) { ty0 = NS_INS; ln0 = L; ps0 = pos + 1; n_ev = 1; } }
            else if (type == NS_MIS) {
                if (!L) { ty0 = NS_MIS; ln0 = step; n_ev = 1; }
                else {
                    ty0 = NS_MIS; ln0 = 1; ty1 = NS_INS; ln1 = L; n_ev = 2;
                    if (step - 1 > L) { ty2 = NS_MIS; ln2 = step - 1 - L; n_ev = 3; }
                }
            } else {
                if (!L) { ty0 = NS_DEL; ln0 = step; n_ev = 1; }
                else {
                    ty0 = NS_DEL; ln0 = step > L ? step - L : 1u; n_ev = 1;
                    if (L > step - 1) { ty1 = NS_INS; ln1 = L - (step - 1); n_ev = 2; }
                }
            }
        }
        bool bad = ln0 > NS_EV_LEN_MAX || ln1 > NS_EV_LEN_MAX || ln2 > NS_EV_LEN_MAX;       // (a merged insertion of > 4095 bases)
        ln0 = min(ln0, NS_EV_LEN_MAX); ln1 = min(ln1, NS_EV_LEN_MAX); ln2 = min(ln2, NS_EV_LEN_MAX);
        // one prefix sum for the deleted bases (< 2^24 per block) and the event counts (<= 3 per lane).  The events of a non-insertion
        // iteration change the length by L - del (every case above), so the cumulative shift in front of them follows from the
        // insertion and deletion sums: no prefix sum of its own
        auto dsh = [](uint32_t ty, uint32_t ln) { return ty == NS_INS ? ln : ty == NS_DEL ? 0u - ln : 0u; };
        const uint32_t d0 = n_ev > 0 ? dsh(ty0, ln0) : 0u, d1 = n_ev > 1 ? dsh(ty1, ln1) : 0u;
        const uint32_t dn_incl = wave_incl_scan(del | n_ev << 24);
        const uint32_t del_incl = dn_incl & 0xffffffu, n_incl = dn_incl >> 24;
        const uint32_t slot = s.n + n_incl - n_ev;
        const uint32_t sh = (uint32_t)s.shift + (pend0 + ins_incl - L) - (del_incl - del);       // (used by non-insertion lanes only)
        // (the shift in FRONT of every event must fit its field, as ev_push32 checks it)
        if ((n_ev > 0 && !ev_shift_fits((int32_t)sh)) || (n_ev > 1 && !ev_shift_fits((int32_t)(sh + d0))) || (n_ev > 2 && !ev_shift_fits((int32_t)(sh + d0 + d1)))) bad = true;
        if (__ballot(bad)) s.range = true;
        if (n_ev > 0) { if (slot < s.cap) { ns_event e; e.pos = ps0; e.info = ns_ev_pack(ln0, ty0, (int32_t)sh); s.ev[slot] = e; } }
        if (n_ev > 1) { if (slot + 1 < s.cap) { ns_event e; e.pos = ps1; e.info = ns_ev_pack(ln1, ty1, (int32_t)(sh + d0)); s.ev[slot + 1] = e; } }
        if (n_ev > 2) { if (slot + 2 < s.cap) { ns_event e; e.pos = ps2; e.info = ns_ev_pack(ln2, ty2, (int32_t)(sh + d0 + d1)); s.ev[slot + 2] = e; } }
        const uint32_t n_blk = (uint32_t)__builtin_amdgcn_readlane((int)n_incl, 63);
        if (s.n + n_blk > s.cap) s.overflow = true;
        s.n += n_blk;
        const uint32_t ins_tot = (uint32_t)__builtin_amdgcn_readlane((int)ins_incl, 63);
        const uint32_t del_tot = (uint32_t)__builtin_amdgcn_readlane((int)del_incl, 63);
        l_new += (int32_t)ins_tot - (int32_t)del_tot;                                                      // S:1808-1815, 1820
        // pending insertion behind the last non-insertion iteration of the block
        const uint32_t pend1 = NB ? ins_tot - (uint32_t)__shfl((int)ins_incl, 63 - __clzll((long long)NB)) : pend0 + ins_tot;
        s.shift = (int32_t)((uint32_t)s.shift + (pend0 + ins_tot - pend1) - del_tot);                     // the insertions filed - the deletions
        pend0 = pend1;
        pos0 += (uint32_t)__builtin_amdgcn_readlane((int)adv_incl, (int)(n_exec - 1u));                   // positions advanced by the iterations that ran
        if (n_exec < 64 || pos0 >= (uint32_t)m_ref) break;
    }
    if ((int32_t)pos0 > middle_ref) { l_new += (int32_t)pos0 - middle_ref; middle_ref = (int32_t)pos0; }      // S:1826-1828
    return EList32{l_new, middle_ref};
}

// ---- the same with K consecutive iterations per lane (a round = 64 * K iterations) -------------------------------------------------
// One iteration per lane leaves a wavefront with ~460 instructions in one long dependent chain per block (Philox, the table walk, three
// prefix sums, two cross-lane reads: 3 us per block alone, 7 us next to others — 13 % of the issue slots used, 50 000 reads in 2.6 ms:
// round 6, profiles/r06/ucoop_pmc.log).  With K iterations per lane the K Philox blocks and table walks are independent instruction
// streams, the prefix sums run once per round over the lanes' totals, and the state that crosses lanes is a fold of four numbers
// per lane: {positions advanced, insertions drawn, whether it has a non-insertion iteration, the insertions behind its last one}.
// Same events, same order, same sink state as coop_unaligned_error_list.
template <int K>
__device__ inline EList32 coopk_unaligned_error_list(const Tabs &T, const ChainTab &c, int32_t m_ref, const ns_key &key, uint32_t seg,
                                                     uint32_t attempt, EvSink32 &s, uint32_t lane) {
    int32_t l_new = m_ref, middle_ref = m_ref;
    if (m_ref <= 0) return EList32{l_new, middle_ref};
    uint32_t pos0 = 0, pend0 = 0;                                // position / pending insertion length in front of the round
    auto draw = [&](uint32_t it, int &type, uint32_t &step) {    // what iteration `it` does (S:1787, 1799-1818)
        const u32x4 w = ns_draw(key, ST_UEVENT, seg, attempt, it, 0);
        const uint64_t ut = w.x;
        type = (ut < ns_thr_lt(0.4)) ? 3 : (ut < ns_thr_lt(0.7)) ? NS_MIS : (ut < ns_thr_lt(0.85)) ? NS_INS : NS_DEL;
        step = 1;
        if (type != 3) step = (uint32_t)run_length_t(T, c, type, w.y, w.z);
    };
    int type_n[K]; uint32_t step_n[K];
    #pragma unroll
    for (int k = 0; k < K; ++k) draw(lane * K + k, type_n[k], step_n[k]);
    bool bad = false;
    for (uint32_t it0 = 0;; it0 += 64u * K) {
        int type[K]; uint32_t step[K];
        #pragma unroll
        for (int k = 0; k < K; ++k) { type[k] = type_n[k]; step[k] = step_n[k]; }
        #pragma unroll
        for (int k = 0; k < K; ++k) draw(it0 + 64u * K + lane * K + k, type_n[k], step_n[k]);     // the next round's draws run under this round's sums
        // positions: lane-local prefix + one prefix sum over the lanes' totals
        uint32_t a_loc[K], a_lane = 0;
        #pragma unroll
        for (int k = 0; k < K; ++k) { a_loc[k] = a_lane; a_lane += type[k] == NS_INS ? 0u : step[k]; }
        const uint32_t a_incl = wave_incl_scan(a_lane);
        const uint32_t pos_l = pos0 + a_incl - a_lane;
        // the iterations that still run (a prefix of the round), their insertions / deletions; within the lane: the insertions pending in
        // front of every non-insertion iteration since the previous one of the LANE (L_loc), whether it is the lane's first (`first`)
        bool exec[K], nonins[K], first[K];
        uint32_t L_loc[K], i_loc[K], d_loc[K], run = 0, i_lane = 0, d_lane = 0, a_exec = 0, n_exec_l = 0;
        bool seen = false;
        #pragma unroll
        for (int k = 0; k < K; ++k) {
            exec[k] = pos_l + a_loc[k] < (uint32_t)m_ref;
            if (!exec[k]) { step[k] = 0; type[k] = 3; }
            nonins[k] = exec[k] && type[k] != NS_INS;
            const uint32_t ins = (exec[k] && type[k] == NS_INS) ? step[k] : 0u, del = (exec[k] && type[k] == NS_DEL) ? step[k] : 0u;
            i_loc[k] = i_lane; d_loc[k] = d_lane;
            i_lane += ins; d_lane += del;
            run += ins;
            L_loc[k] = nonins[k] ? run : 0u;
            first[k] = nonins[k] && !seen;
            if (nonins[k]) { run = 0; seen = true; }
            a_exec += (exec[k] && type[k] != NS_INS) ? step[k] : 0u;
            n_exec_l += exec[k] ? 1u : 0u;
        }
        const uint32_t i_incl = wave_incl_scan(i_lane);
        const uint64_t HAS = __ballot(seen);
        const uint64_t below = HAS & ((1ull << lane) - 1ull);
        const int jb = below ? 63 - __clzll((long long)below) : (int)lane;
        const uint32_t tail_b = (uint32_t)__shfl((int)run, jb), incl_b = (uint32_t)__shfl((int)i_incl, jb);
        const uint32_t pend_l = below ? tail_b + (i_incl - i_lane) - incl_b : pend0 + (i_incl - i_lane);       // pending in front of this lane
        // the events of every iteration, at most three each
        uint32_t ty[K][3], ln[K][3], ps[K][3], n_ev[K], n_loc[K], L[K], n_lane = 0;
        #pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint32_t pos = pos_l + a_loc[k], st = step[k];
            L[k] = nonins[k] ? L_loc[k] + (first[k] ? pend_l : 0u) : 0u;
            const uint32_t Lk = L[k];
            ty[k][0] = ty[k][1] = ty[k][2] = 0; ln[k][0] = ln[k][1] = ln[k][2] = 0; ps[k][0] = pos; ps[k][1] = ps[k][2] = pos + 1;
            uint32_t ne = 0;
            if (nonins[k]) {
                if (type[k] == 3) { if (Lk) { ty[k][0] = NS_INS; ln[k][0] = Lk; ps[k][0] = pos + 1; ne = 1; } }
                else if (type[k] == NS_MIS) {
                    if (!Lk) { ty[k][0] = NS_MIS; ln[k][0] = st; ne = 1; }
                    else {
                        ty[k][0] = NS_MIS; ln[k][0] = 1; ty[k][1] = NS_INS; ln[k][1] = Lk; ne = 2;
                        if (st - 1 > Lk) { ty[k][2] = NS_MIS; ln[k][2] = st - 1 - Lk; ne = 3; }
                    }
                } else {
                    if (!Lk) { ty[k][0] = NS_DEL; ln[k][0] = st; ne = 1; }
                    else {
                        ty[k][0] = NS_DEL; ln[k][0] = st > Lk ? st - Lk : 1u; ne = 1;
                        if (Lk > st - 1) { ty[k][1] = NS_INS; ln[k][1] = Lk - (st - 1); ne = 2; }
                    }
                }
            }
            bad |= ln[k][0] > NS_EV_LEN_MAX || ln[k][1] > NS_EV_LEN_MAX || ln[k][2] > NS_EV_LEN_MAX;      // (a merged insertion of > 4095 bases)
            ln[k][0] = min(ln[k][0], NS_EV_LEN_MAX); ln[k][1] = min(ln[k][1], NS_EV_LEN_MAX); ln[k][2] = min(ln[k][2], NS_EV_LEN_MAX);
            n_ev[k] = ne; n_loc[k] = n_lane; n_lane += ne;
        }
        // deleted bases; event counts (<= 3 K per lane) and the iterations that ran (<= K) in one prefix sum
        const uint32_t d_incl = wave_incl_scan(d_lane);
        const uint32_t ne_incl = wave_incl_scan(n_lane | n_exec_l << 16);
        const uint32_t d_front = d_incl - d_lane, n_front = (ne_incl & 0xffffu) - n_lane;
        auto dsh = [](uint32_t t, uint32_t l) { return t == NS_INS ? l : t == NS_DEL ? 0u - l : 0u; };
        #pragma unroll
        for (int k = 0; k < K; ++k) {
            if (!n_ev[k]) continue;
            // the shift in front of the iteration's events: the insertions filed so far - the deletions so far
            const uint32_t sh = (uint32_t)s.shift + (pend0 + (i_incl - i_lane) + i_loc[k] - L[k]) - (d_front + d_loc[k]);
            const uint32_t d0 = dsh(ty[k][0], ln[k][0]), d1 = n_ev[k] > 1 ? dsh(ty[k][1], ln[k][1]) : 0u;
            if (!ev_shift_fits((int32_t)sh) || (n_ev[k] > 1 && !ev_shift_fits((int32_t)(sh + d0))) || (n_ev[k] > 2 && !ev_shift_fits((int32_t)(sh + d0 + d1)))) bad = true;
            const uint32_t slot = s.n + n_front + n_loc[k];
            if (slot < s.cap) { ns_event e; e.pos = ps[k][0]; e.info = ns_ev_pack(ln[k][0], ty[k][0], (int32_t)sh); s.ev[slot] = e; }
            if (n_ev[k] > 1 && slot + 1 < s.cap) { ns_event e; e.pos = ps[k][1]; e.info = ns_ev_pack(ln[k][1], ty[k][1], (int32_t)(sh + d0)); s.ev[slot + 1] = e; }
            if (n_ev[k] > 2 && slot + 2 < s.cap) { ns_event e; e.pos = ps[k][2]; e.info = ns_ev_pack(ln[k][2], ty[k][2], (int32_t)(sh + d0 + d1)); s.ev[slot + 2] = e; }
        }
        // ---- the state behind the round
        const uint32_t ne_tot = (uint32_t)__builtin_amdgcn_readlane((int)ne_incl, 63);
        const uint32_t n_blk = ne_tot & 0xffffu, n_ran = ne_tot >> 16;
        const uint32_t del_tot = (uint32_t)__builtin_amdgcn_readlane((int)d_incl, 63), ins_tot = (uint32_t)__builtin_amdgcn_readlane((int)i_incl, 63);
        // positions advanced by the iterations that ran: all of the round's, except in the round where the loop ends
        uint32_t adv_tot = (uint32_t)__builtin_amdgcn_readlane((int)a_incl, 63);
        if (n_ran < 64u * K) adv_tot = (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan(a_exec), 63);
        if (s.n + n_blk > s.cap) s.overflow = true;
        s.n += n_blk;
        l_new += (int32_t)ins_tot - (int32_t)del_tot;                                                          // S:1808-1815, 1820
        uint32_t pend1 = pend0 + ins_tot;                          // pending behind the last non-insertion iteration of the round
        if (HAS) {
            const int jl = 63 - __clzll((long long)HAS);
            pend1 = (uint32_t)__shfl((int)run, jl) + ins_tot - (uint32_t)__shfl((int)i_incl, jl);
        }
        s.shift = (int32_t)((uint32_t)s.shift + (pend0 + ins_tot - pend1) - del_tot);
        pend0 = pend1;
        pos0 += adv_tot;
        if (n_ran < 64u * K || pos0 >= (uint32_t)m_ref) break;
    }
    if (__ballot(bad)) s.range = true;
    if ((int32_t)pos0 > middle_ref) { l_new += (int32_t)pos0 - middle_ref; middle_ref = (int32_t)pos0; }      // S:1826-1828
    return EList32{l_new, middle_ref};
}

// ---- cooperative error_list: one read per wavefront, for the few longest reads of a batch ---------------------
// The chain is sequential, but what an iteration draws depends on very little state: the error type on the Markov
// state (7 rows), the run length on the error type (3), the next match length on the bin of the previous match
// length (<= 16 columns).  Each lane therefore evaluates ONE future iteration for EVERY possible state / type / bin
// (its Philox block and all table searches, including the fp64 interpolation) and leaves the results in LDS; the
// dependent part that remains is a walk over 64 iterations of four small table reads each (~10x shorter critical
// path than the thread-per-read chain).  Produces exactly the events of chain_error_list.
#define COOP_MAX_BINS 16u
struct CoopLds {
    uint32_t err_tab[64];                 // 2 bits per Markov state
    uint16_t step_tab[64][4];             // run length per error type
    uint32_t match_tab[64][COOP_MAX_BINS];   // next match length | the bin IT falls into << 16 (what the iteration after needs: round 6 — the bin used
                                          // to come from a global-memory byte table inside the sequential walk, one dependent read per event)
    ns_event ev_buf[64];
};

// TM: the front of the blob (transition rows, run-length tables: n_words_mix words) — the wavefront's LDS copy when the launch gave it room
__device__ inline EList32 coop_error_list(const Tabs &TM, const Tabs &T, const ChainTab &c, int32_t m_ref, const ns_key &key, uint32_t seg,
                                          uint32_t attempt, EvSink32 &s, CoopLds &S, uint32_t lane) {
    int32_t l_new = m_ref, pos = 0, middle_ref = m_ref;
    int state = NS_ST_START;
    u32x4 w = ns_draw(key, ST_EVENT, seg, attempt, 0, 0);
    int32_t prev_match = ecdf_lookup_g(T.d(c.fm_hi), T.d(c.fm_vhi), c.fm_n, c.fm_vlo0, T.h(c.fm_guide), w.x);   // S:1843-1850
    if (prev_match < 2) prev_match = 2;
    pos += prev_match;
    uint32_t it0 = 1;
    int32_t last_ins_pos = -1;
    const uint64_t *trans = TM.q(c.trans);
    const int32_t *bins = T.i(c.mm_bin);
    const uint32_t *seg_off = T.u(c.mm_seg_off);
    const uint8_t *bin_lut = reinterpret_cast<const uint8_t *>(T.w + c.mm_bin_lut);
    auto bin_of = [&](int32_t v) -> uint32_t {                                                     // S:1891-1893
        if ((uint32_t)v < 256u) return bin_lut[v];
        uint32_t b = 0;
        for (; b < c.mm_nbins; ++b)
            if (bins[2 * b] <= v && v < bins[2 * b + 1]) break;
        return b >= c.mm_nbins ? c.mm_nbins - 1 : b;
    };
    uint32_t pbin = bin_of(prev_match);                                                            // bin of the previous match (wave-uniform)
    while (pos < middle_ref) {                                                                     // S:1858
        // ---- parallel: lane evaluates iteration it0+lane for every state / type / bin
        {
            const u32x4 wi = ns_draw(key, ST_EVENT, seg, attempt, it0 + lane, 0);
            uint32_t eb = 0;
#pragma unroll
            for (int st = 0; st < 7; ++st) eb |= (uint32_t)trans_pick_u(trans + 3 * st, wi.x) << (2 * st);
            S.err_tab[lane] = eb;
#pragma unroll
            for (int t = 0; t < 3; ++t) S.step_tab[lane][t] = (uint16_t)run_length_t(TM, c, t, wi.y, wi.z);
            if (c.int_image) {
                // Round 6: the next match length of EVERY column on the one-word segments of the full columns, four columns at a time, their
                // reads side by side: 65 536-cell guide -> the guide's segment and the next -> (unit-wide, no third segment in the cell:
                // all but a few draws in ten thousand) the value.  Two dependent global reads per group of four columns; the fp64 loop below
                // made two to four per COLUMN, one column after the other — and the chain phase of a call waits for the wave that walks
                // its longest read (profiles/r06/chain_trained_shape.log).  Same thresholds, same values as ecdf_lookup_g.
                const uint16_t *g16 = T.h(c.mm_g16);
                const uint64_t *gvf = T.q(c.mm_gv_full);
                const uint32_t cell = wi.w >> 16, nb = c.mm_nbins;
                const uint64_t uu = wi.w;
                for (uint32_t b0 = 0; b0 < nb; b0 += 4) {
                    uint32_t o[4], nc[4], s0[4];
                    uint64_t ga[4], gb[4];
#pragma unroll
                    for (uint32_t k = 0; k < 4; ++k) {
                        const uint32_t b = min(b0 + k, nb - 1u);
                        o[k] = seg_off[b]; nc[k] = seg_off[b + 1] - o[k];
                        s0[k] = g16[(size_t)b * 65536u + cell];
                    }
#pragma unroll
                    for (uint32_t k = 0; k < 4; ++k) {
                        const uint32_t sc = min(s0[k], nc[k] ? nc[k] - 1u : 0u);
                        ga[k] = gvf[o[k] + sc]; gb[k] = gvf[o[k] + sc + 1u];                       // (the word behind a column is read, never used)
                    }
#pragma unroll
                    for (uint32_t k = 0; k < 4; ++k) {
                        const uint32_t b = min(b0 + k, nb - 1u);
                        const uint32_t k0 = (s0[k] < nc[k] ? 1u : 0u) & (uu >= NS_G_THR(ga[k]) ? 1u : 0u);
                        const uint32_t k1 = k0 & (s0[k] + 1u < nc[k] ? 1u : 0u) & (uu >= NS_G_THR(gb[k]) ? 1u : 0u);
                        const uint64_t g = k0 ? gb[k] : ga[k];
                        int32_t v = (int32_t)(uint32_t)(g >> 35) - 1;
                        if (!((s0[k] + k0 < nc[k]) && !k1 && (g & NS_GV_UNIT)))
                            v = ecdf_lookup_gv(gvf + o[k], nc[k], T.h(c.mm_guide) + 256u * b, wi.w, T.q(c.sub2_full), T.d(c.mm_hi) + o[k], T.d(c.mm_vhi) + o[k], T.d(c.mm_vlo0)[b]);
                        // (the value as the walk will use it — "no two 0-matches", S:1900-1901, is decided there — and the bin of either outcome)
                        if (b0 + k < nb) S.match_tab[lane][b0 + k] = (uint32_t)(uint16_t)v | bin_of(v) << 16 | bin_of(v == 0 ? 1 : v) << 24;
                    }
                }
            } else
            for (uint32_t b = 0; b < c.mm_nbins; ++b) {
                const uint32_t o = seg_off[b];
                const int32_t v = ecdf_lookup_g(T.d(c.mm_hi) + o, T.d(c.mm_vhi) + o, seg_off[b + 1] - o, T.d(c.mm_vlo0)[b], T.h(c.mm_guide) + 256 * b, wi.w);
                S.match_tab[lane][b] = (uint32_t)(uint16_t)v | bin_of(v) << 16 | bin_of(v == 0 ? 1 : v) << 24;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- sequential walk (uniform over the wavefront): four small LDS reads per iteration
        uint32_t nl = 0;                                           // events buffered in this block
        uint32_t i = 0;
        for (; i < 64 && pos < middle_ref; ++i) {
            const int error = (int)((S.err_tab[i] >> (2 * state)) & 3u);                          // S:1860-1864
            int32_t step = S.step_tab[i][error];                                                  // S:1866-1873
            if (error == NS_INS) l_new += step; else if (error == NS_DEL) l_new -= step;
            int32_t epos = pos;
            if (error != NS_INS) {                                                                 // S:1875-1880
                pos += step;
                if (pos >= middle_ref) { l_new += pos - middle_ref; middle_ref = pos; }
            } else {                                                                               // S:1881-1882
                if (last_ins_pos == pos && (nl > 0 || s.n > 0)) {                                  // dict key collision
                    if (nl > 0) --nl; else --s.n;
                    s.shift -= (int32_t)s.last_ins_len;
                }
                last_ins_pos = pos;
            }
            {   // ev_push32 into the block buffer
                const uint32_t l = step > (int32_t)NS_EV_LEN_MAX ? NS_EV_LEN_MAX : (uint32_t)step;
                if (step > (int32_t)NS_EV_LEN_MAX || !ev_shift_fits(s.shift)) s.range = true;
                ns_event e; e.pos = (uint32_t)epos; e.info = ns_ev_pack(l, (uint32_t)error, s.shift);
                if (lane == 0) S.ev_buf[nl] = e;
                ++nl;
                if (error == NS_INS) { s.shift += (int32_t)l; s.last_ins_len = l; } else if (error == NS_DEL) s.shift -= (int32_t)l;
            }
            state = NS_ST_MIS + error;                                                             // S:1884
            const uint32_t me = S.match_tab[i][pbin];                                              // S:1891-1898: the column of the previous match's bin
            step = (int32_t)(me & 0xffffu);
            pbin = (me >> 16) & 0xffu;
            if (prev_match == 0 && step == 0) { step = 1; pbin = me >> 24; }                       // S:1900-1901
            prev_match = step;
            if (pos + prev_match > middle_ref) { l_new += pos + prev_match - middle_ref; middle_ref = pos + prev_match; }
            pos += prev_match;
            if (prev_match == 0) state += 3;                                                       // S:1913-1914
            else last_ins_pos = -1;
        }
        // ---- flush the block's events, one lane per event
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane < nl) {
            if (s.n + lane < s.cap) s.ev[s.n + lane] = S.ev_buf[lane];
        }
        if (s.n + nl > s.cap) s.overflow = true;
        s.n += nl;
        it0 += i;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    return EList32{l_new, middle_ref};
}
