// ns_materialise.h — the record kernels' device code: one read per wavefront, tiles of <= 2 KB of output whose 16-byte chunks are
// aligned in the destination.  Per tile (materialise_piece7 below has the details):
//   1. the events that start inside the tile (<= 63; lane = event, prefetched with their letter words) are staged in LDS;
//   2. lane per 16-byte chunk: histogram + wavefront prefix maximum -> the event in force at the chunk's first byte; the bytes copied
//      under it come with ONE unaligned 16-byte global load straight from the source (no source tile in LDS: neighbouring lanes hit the
//      same lines in L1/L2) and stay in the lane's registers;
//   3. lane per event: the bytes copied behind an event that starts INSIDE a chunk (one more such load, masked) and the substituted /
//      inserted letters (mutate_read, S:1965-1995) go to an LDS tile at their output offsets;
//   4. lane per chunk again: own bytes | what the LDS tile holds, IUPAC codes resolved (case_convert, S:743-755), FASTQ: the quality class
//      of every base leaves as two bits for k_qualities, complement / reverse in registers (S:1433-1435, 1675-1680), one aligned 16-byte
//      nontemporal store per line.
// The SOURCE of a piece is the reference (MAT_REF), or — second pass of -k — the pre-homopolymer read in the scratch buffer with
// the homopolymer edits as its event list (MAT_HP_FINAL).  Tiles that straddle the origin of a circular chromosome take the generic
// per-byte path (slow_piece_range).
#pragma once
#include "ns_device.h"

#ifndef NS_TILE_CHUNKS
#define NS_TILE_CHUNKS 2u                                  // 16-byte chunks per lane and tile
#endif
#define T_OUT (1024u * NS_TILE_CHUNKS)
#define T_EV 64u            // events staged per tile: slot 0 = the event in force at the tile start, slots 1..63 = lanes 0..62
#define T_DUMP (T_OUT + 16u)
// quality class of an emitted base travels in two spare bits of its ASCII code (A 41, C 43, G 47, T 54: bits 3 and 5 are free)
// until the qualities are drawn: bit 3 = substituted ('mis'), bit 5 = inserted ('ins', the base is in lower case)
#define NS_CLS_MIS_BIT 0x08u
#define NS_CLS_INS_BIT 0x20u
#define NS_CLS_STRIP 0xd7d7d7d7u
// LDS copy of the quality bucket tables: slots 0..2 = match / mis / ins, slot 3 = unmapped (gaps of chimeric reads)
#define NS_QLUT_SLOTS 5u
// inclusive prefix sum over the wavefront with DPP row shifts / row broadcasts: 6 VALU instructions — with bound_ctrl on the row shifts
// (a lane without a source reads 0) every step folds into ONE v_add_u32_dpp; with bound_ctrl off the compiler emitted v_mov + v_mov_dpp +
// v_add per step (18 instead of 6: round 6, the ISA of k_hp_scan)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);       // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);       // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);       // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);       // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);      // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);      // row_bcast:31 -> rows 2, 3
    return v;
}

// inclusive prefix maximum over the wavefront (values >= 0), same DPP pattern
__device__ __forceinline__ uint32_t wave_incl_max(uint32_t v) {
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));
    return v;
}
__device__ __forceinline__ uint32_t dpp_wave_shl1(uint32_t lane63_value, uint32_t v) {     // lane l gets v of lane l + 1, lane 63 gets lane63_value
    return (uint32_t)__builtin_amdgcn_update_dpp((int)lane63_value, (int)v, 0x130, 0xf, 0xf, false);
}

// wave-uniform values that reach the kernel through vector loads: pin them to SGPRs so that the arithmetic on them is scalar
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t uni64(uint64_t v) { return (uint64_t)uni((uint32_t)v) | (uint64_t)uni((uint32_t)(v >> 32)) << 32; }

struct ReadOut {
    uint8_t *seq;        // first base of the record's sequence line
    uint8_t *qual;       // first quality character, or nullptr
    uint32_t seq_len;
    bool reversed;
    bool uracil;         // --uracil: T -> U in the emitted bases (S:1247-1248)
};
// 'T' (0x54) -> 'U' (0x55) on 8 packed bytes: exact zero-byte detect of x ^ 0x54..54
__device__ __forceinline__ uint64_t t_to_u8(uint64_t x) {
    const uint64_t t = x ^ 0x5454545454545454ull;
    const uint64_t z = ~(((t & 0x7f7f7f7f7f7f7f7full) + 0x7f7f7f7f7f7f7f7full) | t) & 0x8080808080808080ull;
    return x | (z >> 7);
}

// A full group leaves as a NONTEMPORAL store (round 5): the record image, the scratch image of -k and the error profile are streams
// that are written once, wavefront-wide and contiguous, and not read back by the kernel that writes them — kept out of the L2 they no
// longer evict the reference and the event lists the same kernel is reading (same box, profiles/r05/ab_nt_records.log: record kernel
// 5.43 -> 5.09 ms, whole step 9.64 -> 9.27 ms; k_errlog 7.64 -> 6.34 ms, ab_errlog_nt.log)
typedef uint32_t ns_v4u_any __attribute__((ext_vector_type(4), aligned(1)));      // 16 bytes at any address
__device__ __forceinline__ void store16(uint8_t *dst, uint32_t count, uint64_t lo, uint64_t hi) {
    if (count == 16) {
        ns_v4u_any x = {(uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32)};
        __builtin_nontemporal_store(x, reinterpret_cast<ns_v4u_any *>(dst));
    } else {                             // 8 + 4 + 2 + 1: at most four (unaligned) stores instead of up to 15 byte stores
        if (count & 8u) { __builtin_memcpy(dst, &lo, 8); dst += 8; lo = hi; }
        if (count & 4u) { const uint32_t w = (uint32_t)lo; __builtin_memcpy(dst, &w, 4); dst += 4; lo >>= 32; }
        if (count & 2u) { const uint16_t w = (uint16_t)lo; __builtin_memcpy(dst, &w, 2); dst += 2; lo >>= 16; }
        if (count & 1u) *dst = (uint8_t)lo;
    }
}
// bytes [i0, 16) of a 16-byte group moved down to byte 0
__device__ __forceinline__ void shift_down_bytes(uint64_t &lo, uint64_t &hi, uint32_t i0) {
    const uint32_t sh = 8 * i0;
    if (sh == 0) return;
    if (sh < 64) { lo = (lo >> sh) | (hi << (64 - sh)); hi >>= sh; }
    else { lo = hi >> (sh - 64); hi = 0; }
}
__device__ __forceinline__ void reverse_bytes(uint64_t &lo, uint64_t &hi, uint32_t count) {
    uint64_t rlo = __builtin_bswap64(hi), rhi = __builtin_bswap64(lo);     // byte i -> 15-i
    uint32_t s = 8 * (16 - count);
    if (s == 0) { lo = rlo; hi = rhi; }
    else if (s < 64) { lo = (rlo >> s) | (rhi << (64 - s)); hi = rhi >> s; }
    else { lo = rhi >> (s - 64); hi = 0; }
}
__device__ __forceinline__ uint64_t complement8(uint64_t x) {              // A<->T, C<->G on 8 packed ASCII bases
    return x ^ 0x1515151515151515ull ^ (((x >> 1) & 0x0101010101010101ull) * 0x11ull);
}
// `count` bytes whose pre-revcomp coordinates are [q0, q0+count), byte i in bits 8i of (lo,hi)
__device__ __forceinline__ void store_chunk(const ReadOut &ro, uint32_t q0, uint32_t count, uint64_t lo, uint64_t hi,
                                            uint64_t qlo, uint64_t qhi, bool ascii_quals = false) {
    uint32_t o0 = q0;
    if (ro.reversed) {
        lo = complement8(lo); hi = complement8(hi);
        reverse_bytes(lo, hi, count);
        o0 = ro.seq_len - q0 - count;
    }
    if (ro.uracil) { lo = t_to_u8(lo); hi = t_to_u8(hi); }
    store16(ro.seq + o0, count, lo, hi);
    if (ro.qual) {
        if (!ascii_quals) { qlo += 0x2121212121212121ull; qhi += 0x2121212121212121ull; }        // chr(q + 33), S:1441
        if (ro.reversed) reverse_bytes(qlo, qhi, count);
        store16(ro.qual + o0, count, qlo, qhi);
    }
}
// the same in two steps: final bytes / record offset now, the store later (k_materialise keeps one chunk per lane pending so
// that the store is issued behind the next tile's loads and never sits in front of a load the wave waits for)
struct PendingChunk { uint64_t lo, hi, qlo, qhi; uint32_t o0, count; };
__device__ __forceinline__ PendingChunk prep_chunk(const ReadOut &ro, uint32_t q0, uint32_t count, uint64_t lo, uint64_t hi,
                                                   uint64_t qlo, uint64_t qhi) {
    PendingChunk p; p.count = count; p.o0 = q0;
    if (ro.reversed) {
        if (count == 16) {              // full group: complement through a byte-permute look-up (A 1, C 3, T 4, G 7 after & 7), then reverse
            const uint32_t w0 = (uint32_t)lo, w1 = (uint32_t)(lo >> 32), w2 = (uint32_t)hi, w3 = (uint32_t)(hi >> 32);
            const uint32_t c0 = __builtin_amdgcn_perm(0x43000041u, 0x47005400u, w0 & 0x07070707u), c1 = __builtin_amdgcn_perm(0x43000041u, 0x47005400u, w1 & 0x07070707u);
            const uint32_t c2 = __builtin_amdgcn_perm(0x43000041u, 0x47005400u, w2 & 0x07070707u), c3 = __builtin_amdgcn_perm(0x43000041u, 0x47005400u, w3 & 0x07070707u);
            const uint32_t r0 = __builtin_amdgcn_perm(0u, c3, 0x00010203u), r1 = __builtin_amdgcn_perm(0u, c2, 0x00010203u);
            const uint32_t r2 = __builtin_amdgcn_perm(0u, c1, 0x00010203u), r3 = __builtin_amdgcn_perm(0u, c0, 0x00010203u);
            lo = (uint64_t)r0 | (uint64_t)r1 << 32; hi = (uint64_t)r2 | (uint64_t)r3 << 32;
        } else {
            lo = complement8(lo); hi = complement8(hi);
            reverse_bytes(lo, hi, count);
        }
        p.o0 = ro.seq_len - q0 - count;
    }
    if (ro.uracil) { lo = t_to_u8(lo); hi = t_to_u8(hi); }
    p.lo = lo; p.hi = hi; p.qlo = 0; p.qhi = 0;
    if (ro.qual) {
        qlo += 0x2121212121212121ull; qhi += 0x2121212121212121ull;         // chr(q + 33), S:1441
        if (ro.reversed) reverse_bytes(qlo, qhi, count);
        p.qlo = qlo; p.qhi = qhi;
    }
    return p;
}
__device__ __forceinline__ void flush_chunk(const ReadOut &ro, PendingChunk &p) {
    if (!p.count) return;
    store16(ro.seq + p.o0, p.count, p.lo, p.hi);
    if (ro.qual) store16(ro.qual + p.o0, p.count, p.qlo, p.qhi);
    p.count = 0;
}
__device__ __forceinline__ void put_byte(uint64_t &lo, uint64_t &hi, uint32_t i, uint32_t b) {
    if (i < 8) lo |= (uint64_t)b << (8 * i); else hi |= (uint64_t)b << (8 * (i - 8));
}

// cached 16-bit quality draws: block = mpos >> 3
struct QualDraw {
    uint32_t blk; u32x4 w;
};
__device__ __forceinline__ uint32_t qual_draw(QualDraw &qd, const DevModel &m, int cls, const ns_key &key, uint32_t stream,
                                              uint32_t seg, uint32_t attempt, uint32_t mpos) {
    if ((mpos >> 3) != qd.blk) { qd.blk = mpos >> 3; qd.w = ns_draw(key, stream, seg, attempt, qd.blk, 0); }
    uint32_t h = (ns_word(qd.w, (mpos & 7) >> 1) >> (16 * (mpos & 1))) & 0xffffu;
    return qual_value_lut(m.qual_thr + cls * NS_QUAL_LEVELS, m.qual_lut + cls * 1024, h);
}

// head / tail: uniform bases (S:1426-1427) + 'ht' qualities (S:1421-1423).  One Philox block per 64 letters.  Both regions in
// one pass: lanes 0..31 take 16-letter groups of the head, lanes 32..63 of the tail (one Philox evaluation for the wavefront).
__device__ inline void emit_head_tail(const DevModel &m, const ReadOut &ro, const ns_key &key, uint32_t a, uint32_t head, uint32_t tail,
                                      uint32_t lane) {
    const bool is_tail = lane >= 32;
    const uint32_t len = is_tail ? tail : head, stream = is_tail ? ST_TAIL : ST_HEAD;
    const uint32_t q_start = is_tail ? ro.seq_len - tail : 0u, hq_off = is_tail ? head : 0u;
    for (uint32_t i0 = (lane & 31u) * 16; i0 < len; i0 += 32 * 16) {
        const uint32_t count = min(16u, len - i0);
        u32x4 w = ns_draw(key, stream, 0, a, i0 >> 6, 0);
        const uint32_t word = ns_word(w, (i0 >> 4) & 3);
        uint32_t l4[4];
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) {                       // letters 4k..4k+3: 2-bit fields -> byte selectors -> "ATCG"
            const uint32_t x = (word >> (8 * k)) & 0xffu;
            const uint32_t t = (x | x << 12) & 0x000f000fu;
            l4[k] = __builtin_amdgcn_perm(0u, 0x47435441u, (t | t << 6) & 0x03030303u);
        }
        uint64_t lo = (uint64_t)l4[0] | (uint64_t)l4[1] << 32, hi = (uint64_t)l4[2] | (uint64_t)l4[3] << 32, qlo = 0, qhi = 0;
        if (ro.qual) {
            QualDraw qd; qd.blk = 0xffffffffu;
            for (uint32_t i = 0; i < count; ++i) put_byte(qlo, qhi, i, qual_draw(qd, m, NS_Q_HT, key, ST_HTQ, 0, a, hq_off + i0 + i));
        }
        store_chunk(ro, q_start + i0, count, lo, hi, qlo, qhi);
    }
}

// polyA tail of a transcriptome read (S:1224-1225): `len` A's after the last piece; their qualities are the LAST `len` values of the
// head/tail quality draw, in reverse order (S:1229-1231: popped from the end and appended)
__device__ inline void emit_polya(const DevModel &m, const ReadOut &ro, const ns_key &key, uint32_t a, uint32_t q_start, uint32_t len,
                                  uint32_t head, uint32_t tail, uint32_t lane) {
    for (uint32_t i0 = lane * 16; i0 < len; i0 += 64 * 16) {
        const uint32_t count = min(16u, len - i0);
        uint64_t qlo = 0, qhi = 0;
        if (ro.qual) {
            QualDraw qd; qd.blk = 0xffffffffu;
            for (uint32_t i = 0; i < count; ++i) put_byte(qlo, qhi, i, qual_draw(qd, m, NS_Q_HT, key, ST_HTQ, 0, a, head + tail + len - 1 - (i0 + i)));
        }
        store_chunk(ro, q_start + i0, count, 0x4141414141414141ull, 0x4141414141414141ull, qlo, qhi);
    }
}

// `count` quality values whose pre-revcomp coordinates are [q0, q0 + count)
__device__ __forceinline__ void store_qual_chunk(const ReadOut &ro, uint32_t q0, uint32_t count, uint64_t qlo, uint64_t qhi) {
    uint32_t o0 = q0;
    qlo += 0x2121212121212121ull; qhi += 0x2121212121212121ull;             // chr(q + 33), S:1441
    if (ro.reversed) {
        if (count == 16) { const uint64_t t = __builtin_bswap64(qhi); qhi = __builtin_bswap64(qlo); qlo = t; }
        else reverse_bytes(qlo, qhi, count);
        o0 = ro.seq_len - q0 - count;
    }
    store16(ro.qual + o0, count, qlo, qhi);
}
// the quality half of emit_polya (k_qualities: the record kernel of a FASTQ batch writes the bases only)
__device__ inline void emit_polya_quals(const DevModel &m, const ReadOut &ro, const ns_key &key, uint32_t a, uint32_t q_start, uint32_t len,
                                        uint32_t head, uint32_t tail, uint32_t lane) {
    for (uint32_t i0 = lane * 16; i0 < len; i0 += 64 * 16) {
        const uint32_t count = min(16u, len - i0);
        uint64_t qlo = 0, qhi = 0;
        QualDraw qd; qd.blk = 0xffffffffu;
        for (uint32_t i = 0; i < count; ++i) put_byte(qlo, qhi, i, qual_draw(qd, m, NS_Q_HT, key, ST_HTQ, 0, a, head + tail + len - 1 - (i0 + i)));
        store_qual_chunk(ro, q_start + i0, count, qlo, qhi);
    }
}

// ---- generic per-byte path (global memory, no staging) ---------------------------------------------------
struct PieceCtx {
    const ns_event *ev;
    const uint32_t *wd;      // -k, second pass: the letter word of every homopolymer edit (k_hp_drain); else unused (event_word)
    uint32_t n_ev;
    uint32_t out_len, ref_len;
    uint64_t chrom_base;
    uint64_t chrom_len;
    uint64_t pos;
    uint32_t sid;
    uint32_t kind;
};
struct Cursor {
    uint32_t j;
    uint32_t cur_out, cur_pl, cur_type, cur_pos, cur_rp;
    uint32_t next_out;
};
__device__ __forceinline__ uint32_t ev_out_start(const ns_event &e) { return (uint32_t)((int32_t)e.pos + ns_ev_shift(e.info)); }
__device__ __forceinline__ void cursor_load(Cursor &c, const PieceCtx &pc) {
    if (c.j == 0) { c.cur_out = 0; c.cur_pl = 0; c.cur_type = 3; c.cur_pos = 0; c.cur_rp = 0; }
    else {
        ns_event e = pc.ev[c.j - 1];
        uint32_t len = ns_ev_len(e.info), ty = ns_ev_type(e.info);
        c.cur_out = ev_out_start(e); c.cur_type = ty; c.cur_pos = e.pos;
        c.cur_pl = (ty == NS_DEL) ? 0 : len;
        c.cur_rp = e.pos + ((ty == NS_INS) ? 0 : len);
    }
    c.next_out = (c.j < pc.n_ev) ? ev_out_start(pc.ev[c.j]) : 0xffffffffu;
}
__device__ __forceinline__ void cursor_seek(Cursor &c, const PieceCtx &pc, uint32_t m) {
    uint32_t lo = 0, hi = pc.n_ev;
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if (ev_out_start(pc.ev[mid]) <= m) lo = mid + 1; else hi = mid;
    }
    c.j = lo;
    cursor_load(c, pc);
}
NS_DEV uint8_t ref_base_at(const DevRef &ref, const PieceCtx &pc, uint32_t x) {
    uint64_t g = pc.pos + x;
    if (g >= pc.chrom_len) g -= pc.chrom_len;               // circular wrap (S:1757-1760)
    return ref.bases[pc.chrom_base + g];
}
__device__ __forceinline__ uint8_t piece_byte(const DevRef &ref, const PieceCtx &pc, Cursor &c, uint32_t m,
                                              const ns_key &key, uint32_t attempt, int &cls) {
    while (m >= c.next_out) { c.j++; cursor_load(c, pc); }
    uint32_t d = m - c.cur_out;
    if (d < c.cur_pl) {
        if (c.cur_type == NS_MIS) {                                                  // S:1965-1978
            cls = NS_Q_MIS;
            uint32_t x = c.cur_pos + d;
            uint8_t cur = resolve_base(ref_base_at(ref, pc, x), key, pc.sid, attempt, x);
            return mis_letter(cur, key, pc.sid, attempt, c.j - 1, d);
        }
        cls = NS_Q_INS;                                                              // S:1986-1995
        return ins_letter(key, pc.sid, attempt, c.j - 1, d);
    }
    cls = NS_Q_MATCH;
    uint32_t x = c.cur_rp + (d - c.cur_pl);
    return resolve_base(ref_base_at(ref, pc, x), key, pc.sid, attempt, x);           // case_convert, S:743-755
}
NS_DEV PieceCtx load_piece(const ns_event *events, const DevRef &ref, const ns_piece &p, uint32_t pi) {
    PieceCtx pc;
    pc.ev = events + p.ev_off; pc.wd = nullptr; pc.n_ev = p.n_ev; pc.out_len = p.out_len; pc.ref_len = p.ref_len;
    pc.chrom_base = ref.chrom_off[p.chrom];
    pc.chrom_len = ref.chrom_off[p.chrom + 1] - pc.chrom_base;
    if (p.ref_gpos >= NS_SPLICED_BASE) {                    // intron retention: the stretch lies in the splice arena; pos is a genome coordinate
        pc.chrom_base = (uint64_t)((uintptr_t)ref.spliced - (uintptr_t)ref.bases) + (p.ref_gpos - NS_SPLICED_BASE) - p.pos;
        pc.chrom_len = ~0ull;
    }
    pc.pos = p.pos; pc.kind = p.kind;
    pc.sid = p.kind ? NS_GAP_SEG + (pi >> 1) : (pi >> 1);
    return pc;
}
// the same for kernels in which the whole wavefront works on one piece: everything wave-uniform, held in SGPRs
__device__ __forceinline__ PieceCtx load_piece_uniform(const ns_event *events, const DevRef &ref, const ns_piece &p, uint32_t pi,
                                                       const uint32_t *ev_word = nullptr) {
    PieceCtx pc;
    const uint64_t eo = uni64(p.ev_off);
    pc.ev = events + eo; pc.wd = ev_word + eo; pc.n_ev = uni(p.n_ev); pc.out_len = uni(p.out_len); pc.ref_len = uni(p.ref_len);
    const uint32_t chrom = uni(p.chrom);
    pc.chrom_base = uni64(ref.chrom_off[chrom]);
    pc.chrom_len = uni64(ref.chrom_off[chrom + 1]) - pc.chrom_base;
    pc.pos = uni(p.pos); pc.kind = uni(p.kind);
    const uint64_t gpos = uni64(p.ref_gpos);
    if (gpos >= NS_SPLICED_BASE) {
        pc.chrom_base = (uint64_t)((uintptr_t)ref.spliced - (uintptr_t)ref.bases) + (gpos - NS_SPLICED_BASE) - pc.pos;
        pc.chrom_len = ~0ull;
    }
    pc.sid = pc.kind ? NS_GAP_SEG + (pi >> 1) : (pi >> 1);
    return pc;
}
// bytes [m_lo, m_hi) of one piece, 16 per lane, straight from global memory.  CLS_BITS: the bases keep their quality class in bits
// 3 / 5 (first record pass of -k with FASTQ: the qualities are drawn by the second pass)
template <bool CLS_BITS = false>
__device__ inline void slow_piece_range(const DevModel &m, const DevRef &ref, const ReadOut &ro, const ns_key &key, uint32_t a,
                                        const PieceCtx &pc, uint32_t pq, uint32_t m_lo, uint32_t m_hi, uint32_t lane) {
    for (uint32_t m0 = m_lo + lane * 16; m0 < m_hi; m0 += 64 * 16) {
        const uint32_t count = min(16u, m_hi - m0);
        Cursor cur;
        cursor_seek(cur, pc, m0);
        uint64_t lo = 0, hi = 0, qlo = 0, qhi = 0;
        QualDraw qd; qd.blk = 0xffffffffu;
        for (uint32_t i = 0; i < count; ++i) {
            int cls;
            uint32_t b = piece_byte(ref, pc, cur, m0 + i, key, a, cls);
            if constexpr (CLS_BITS) b |= cls == NS_Q_MIS ? NS_CLS_MIS_BIT : cls == NS_Q_INS ? NS_CLS_INS_BIT : 0u;
            put_byte(lo, hi, i, b);
            if (ro.qual) put_byte(qlo, qhi, i, qual_draw(qd, m, pc.kind ? NS_Q_UNMAPPED : cls, key, ST_QUAL, pc.sid, a, m0 + i));
        }
        store_chunk(ro, pq + m0, count, lo, hi, qlo, qhi);
    }
}

// every wavefront works on its own LDS tile: DS instructions of one wave execute in issue order, so a compiler-level fence is
// all that is needed between an LDS write and a cross-lane LDS read (no s_barrier, no vmcnt drain)
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// LDS pointers by type: volatile, so that the compiler cannot merge neighbouring narrow stores into a wide one at an address that is not
// a multiple of its size, and in the LDS address space explicitly (volatile accesses are not inferred into it).  A ds_write_b16 / b32 /
// b64 (or read) at such an address costs ~58 LDS cycles per CU against ~7 aligned; a byte store costs ~7 wherever it lands
// (scripts/microbench/lds_align.hip, profiles/r05/microbench_lds_align.log).
typedef __attribute__((address_space(3))) volatile uint8_t *LdsBytes;
typedef __attribute__((address_space(3))) volatile uint32_t *LdsWords;
// the first cn <= 16 bytes of (x0, x1, x2, x3) to d, any alignment, with aligned stores only: byte stores up to the next dword boundary,
// whole dwords, byte stores for the rest
__device__ __forceinline__ void lds_put_bytes(LdsBytes d, uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t cn) {
    const uint32_t head = min(cn, (0u - (uint32_t)(uintptr_t)d) & 3u);
    if (head > 0) d[0] = (uint8_t)x0;
    if (head > 1) d[1] = (uint8_t)(x0 >> 8);
    if (head > 2) d[2] = (uint8_t)(x0 >> 16);
    const uint32_t sh = 8u * head;                       // the value moved down by the bytes that are out
    const uint32_t y0 = __builtin_amdgcn_alignbit(x1, x0, sh), y1 = __builtin_amdgcn_alignbit(x2, x1, sh), y2 = __builtin_amdgcn_alignbit(x3, x2, sh), y3 = x3 >> sh;
    const uint32_t rem = cn - head;
    const LdsWords w = (LdsWords)(d + head);
    if (rem >= 4u) w[0] = y0;
    if (rem >= 8u) w[1] = y1;
    if (rem >= 12u) w[2] = y2;
    if (rem >= 16u) w[3] = y3;
    const uint32_t yt = rem >= 12u ? y3 : rem >= 8u ? y2 : rem >= 4u ? y1 : y0, tail = rem & 3u;
    const LdsBytes e = d + head + (rem & ~3u);
    if (tail > 0) e[0] = (uint8_t)yt;
    if (tail > 1) e[1] = (uint8_t)(yt >> 8);
    if (tail > 2) e[2] = (uint8_t)(yt >> 16);
}

// ---- dense pieces: lane per event -----------------------------------------------------------------------------
// Unaligned reads (S:1482-1549) and the gaps of chimeric reads carry ~0.55 events per base: an event owns two output bytes on
// average — its letters, then the reference bases copied behind them up to the next event.  One lane per event (item 0 = the stretch
// in front of the first event): the lane fetches the <= 32 reference bytes its stretch can need with two unaligned 16-byte loads
// (all of a round's loads are independent: no per-byte chains of dependent loads, which is what the longest read of a batch used to
// spend its time on), writes its bytes into an LDS tile of 1024 output bytes, and the tile leaves with one 16-byte store per lane.
#define NS_DENSE_TILE 1024u
struct __align__(16) DenseLds { uint8_t out[NS_DENSE_TILE + 64]; };      // + a dump slot per lane for predicated-off letter stores

template <bool FASTQ>
__device__ inline void dense_piece(const DevModel &m, const DevRef &ref, DenseLds &S, const ReadOut &ro, const ns_key &key, uint32_t a,
                                   const PieceCtx &pc, uint32_t pq, uint32_t m_lo, uint32_t m_hi, uint32_t lane) {
    if (m_lo & 15u) { slow_piece_range(m, ref, ro, key, a, pc, pq, m_lo, m_hi, lane); return; }   // (the tiles below start on a multiple of 16)
    const uint32_t n_items = pc.n_ev + 1u;                    // item jj > 0 = event jj - 1; item 0 starts at output offset 0, copies from 0
    const uint8_t *seg0 = ref.bases + pc.chrom_base + pc.pos;
    const uint64_t lin = pc.chrom_len - pc.pos;               // segment positions below this lie before the origin of a circular chromosome
    uint32_t jb = 0;                                          // item in force at the tile start: the last one that starts at or before it
    if (m_lo) {
        uint32_t lo = 0, hi = pc.n_ev;                        // events with out_start <= m_lo
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (ev_out_start(pc.ev[mid]) <= m_lo) lo = mid + 1; else hi = mid; }
        jb = uni(lo);
    }
    for (uint32_t M0 = m_lo; M0 < m_hi; M0 += NS_DENSE_TILE) {
        const uint32_t M1 = min(M0 + NS_DENSE_TILE, m_hi);
        uint32_t seen = 0;                                    // items with out_start <= M1 met in this tile
        for (uint32_t jj0 = jb;; jj0 += 64) {
            const uint32_t jj = jj0 + lane;
            const bool valid = jj < n_items;
            uint32_t os = 0, pl = 0, ty = 3, rp = 0, pos = 0, nxt = pc.out_len;
            if (valid) {
                if (jj) {
                    const ns_event e = pc.ev[jj - 1];
                    const uint32_t len = ns_ev_len(e.info);
                    ty = ns_ev_type(e.info); os = ev_out_start(e); pos = e.pos;
                    pl = ty == NS_DEL ? 0u : len; rp = e.pos + (ty == NS_INS ? 0u : len);
                }
                if (jj < pc.n_ev) nxt = ev_out_start(pc.ev[jj]);
            }
            const uint32_t lo = max(os, M0), hi = min(nxt, M1);
            const bool act = valid && lo < hi;
            const uint32_t n_let = act ? min(pl, hi - os) : 0u;                       // letters of the event inside the tile
            const uint32_t c_lo = max(lo, os + pl);
            const uint32_t cn = act && c_lo < hi ? hi - c_lo : 0u;                    // copied bases behind them
            const uint32_t x0 = rp + (c_lo - os - pl);                                // output byte mm <- segment position rp + (mm - os - pl)
            // ---- straight-line path: the letters four at a time (byte permutes, as the tiled kernel), the copied bases from one
            // 16-byte load, stored with aligned LDS stores (lds_put_bytes).  Items cut by the tile border, stretches of > 16 bases, IUPAC codes under
            // the copy and the origin of a circular chromosome take the per-byte walk below.
            const bool fast_l = n_let && os >= M0 && (ty == NS_INS || (uint64_t)pos + n_let <= lin);
            bool fast_c = cn && cn <= 16u && (uint64_t)x0 + 16u <= lin;
            uint4 f = make_uint4(0, 0, 0, 0);
            if (fast_c) __builtin_memcpy(&f, seg0 + x0, 16);
            if ((f.x | f.y | f.z | f.w) & 0x80808080u) fast_c = false;             // IUPAC codes (case_convert, S:743-755)
            {
                uint32_t nl_max = fast_l ? n_let : 0u;
                nl_max = max(nl_max, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)nl_max, 0x111, 0xf, 0xf, false));   // row_shr:1
                nl_max = max(nl_max, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)nl_max, 0x112, 0xf, 0xf, false));
                nl_max = max(nl_max, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)nl_max, 0x114, 0xf, 0xf, false));
                nl_max = max(nl_max, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)nl_max, 0x118, 0xf, 0xf, false));
                const uint32_t m01 = max((uint32_t)__builtin_amdgcn_readlane((int)nl_max, 15), (uint32_t)__builtin_amdgcn_readlane((int)nl_max, 31));
                const uint32_t m23 = max((uint32_t)__builtin_amdgcn_readlane((int)nl_max, 47), (uint32_t)__builtin_amdgcn_readlane((int)nl_max, 63));
                const uint32_t groups = (max(m01, m23) + 3u) >> 2;                   // wave-uniform: letter groups of four
                uint32_t frac = 0;
                const uint32_t o_base = os - M0;
                for (uint32_t g = 0; g < groups; ++g) {
                    const uint32_t i0 = 4u * g;
                    const bool on = fast_l && i0 < n_let;
                    if (!(g & 3u) && on) frac = payload_word(key, pc.sid, a, jj - 1u, g >> 2);   // letter i = field / digit i & 15 of word i >> 4
                    uint32_t cur4 = 0x41414141u;
                    if (on && ty != NS_INS) {
                        __builtin_memcpy(&cur4, seg0 + pos + i0, 4);
                        if (cur4 & 0x80808080u) {                                      // IUPAC code under a substitution: case_convert first
                            uint32_t rs = 0;
                            for (uint32_t t = 0; t < 4u; ++t) rs |= (uint32_t)resolve_base((cur4 >> (8u * t)) & 0xffu, key, pc.sid, a, pos + i0 + t) << (8u * t);
                            cur4 = rs;
                        }
                    }
                    // insertion: 2-bit fields -> "ATCG" (S:1990)
                    const uint32_t x8 = (frac >> (8u * (g & 3u))) & 0xffu, t8 = (x8 | x8 << 12) & 0x000f000fu;
                    const uint32_t ins4 = __builtin_amdgcn_perm(0u, 0x47435441u, (t8 | t8 << 6) & 0x03030303u);
                    // substitution: the next four base-3 digits pick among the three other bases (S:1968-1972)
                    uint32_t f3 = frac;
                    const uint32_t d0 = next_digit3(f3), d1 = next_digit3(f3), d2 = next_digit3(f3), d3 = next_digit3(f3);
                    const uint32_t d4 = d0 | d1 << 8 | d2 << 16 | d3 << 24;
                    const uint32_t vv = (cur4 >> 1) & 0x03030303u;                           // A 0, C 1, T 2, G 3 (N counts as G: no digit reaches its rank)
                    const uint32_t rank4 = (vv & 0x01010101u) << 1 | ((vv >> 1) & 0x01010101u);   // rank in "ATCG": A 0, T 1, C 2, G 3
                    const uint32_t ge = ((d4 | 0x80808080u) - rank4) & 0x80808080u;          // per byte: digit >= rank
                    const uint32_t mis4 = __builtin_amdgcn_perm(0u, 0x47435441u, d4 + (ge >> 7));
                    const bool is_ins = ty == NS_INS;
                    if (!is_ins) frac = f3;
                    const uint32_t letters = is_ins ? ins4 : mis4;
                    const uint32_t cnt = on ? n_let - i0 : 0u, dump = NS_DENSE_TILE + lane, o = o_base + i0;
                    S.out[cnt > 0 ? o : dump] = (uint8_t)letters; S.out[cnt > 1 ? o + 1 : dump] = (uint8_t)(letters >> 8);
                    S.out[cnt > 2 ? o + 2 : dump] = (uint8_t)(letters >> 16); S.out[cnt > 3 ? o + 3 : dump] = (uint8_t)(letters >> 24);
                }
            }
            if (fast_c) lds_put_bytes((LdsBytes)&S.out[c_lo - M0], f.x, f.y, f.z, f.w, cn);
            if (act && ((n_let && !fast_l) || (cn && !fast_c))) {
                // ---- per-byte walk.  Letters (mutate_read, S:1965-1995): letter i = field / digit i & 15 of word i >> 4
                if (!fast_l) {
                    uint32_t word = 0;
                    for (uint32_t i = 0; i < n_let; ++i) {
                        if (!(i & 15u)) word = payload_word(key, pc.sid, a, jj - 1u, i >> 4);
                        uint32_t b;
                        if (ty == NS_INS) b = bases_atcg((word >> (2u * (i & 15u))) & 3u);
                        else {
                            const uint32_t x = pos + i;
                            b = mis_from_digit(resolve_base(ref_base_at(ref, pc, x), key, pc.sid, a, x), next_digit3(word));
                        }
                        if (os + i >= lo) S.out[os + i - M0] = (uint8_t)b;
                    }
                }
                if (cn && !fast_c) {
                    for (uint32_t i = 0; i < cn; ++i) {
                        const uint32_t x = x0 + i;
                        S.out[c_lo + i - M0] = resolve_base(ref_base_at(ref, pc, x), key, pc.sid, a, x);
                    }
                }
            }
            seen += (uint32_t)__popcll(__ballot(valid && os <= M1));
            if (__ballot(valid && os >= M1) || jj0 + 64u >= n_items) break;       // an item behind the tile was met / no items left
        }
        jb = jb + seen - 1u;                                  // the last item that starts at or before M1 is in force there
        wave_sync();
        // ---- the tile leaves: 16 bytes per lane (+ their qualities: class 'unmapped', S:1521, 1564)
        const uint32_t m0 = M0 + 16u * lane;
        if (m0 < M1) {
            const uint32_t count = min(16u, M1 - m0);
            const uint4 v = *reinterpret_cast<const uint4 *>(&S.out[16u * lane]);
            uint64_t qlo = 0, qhi = 0;
            if constexpr (FASTQ) {
                const u32x4 w0 = ns_draw(key, ST_QUAL, pc.sid, a, m0 >> 3, 0), w1 = ns_draw(key, ST_QUAL, pc.sid, a, (m0 >> 3) + 1u, 0);
                const uint32_t cls = pc.kind ? (uint32_t)NS_Q_UNMAPPED : (uint32_t)NS_Q_MATCH;
                const uint32_t *thr = m.qual_thr + cls * NS_QUAL_LEVELS;
                const uint16_t *lut = m.qual_lut + cls * 1024u;
#pragma unroll
                for (uint32_t i = 0; i < 8; ++i) {
                    qlo |= (uint64_t)qual_value_lut(thr, lut, (ns_word(w0, i >> 1) >> (16u * (i & 1u))) & 0xffffu) << (8u * i);
                    qhi |= (uint64_t)qual_value_lut(thr, lut, (ns_word(w1, i >> 1) >> (16u * (i & 1u))) & 0xffffu) << (8u * i);
                }
            }
            store_chunk(ro, pq + m0, count, (uint64_t)v.x | (uint64_t)v.y << 32, (uint64_t)v.z | (uint64_t)v.w << 32, qlo, qhi);
        }
        wave_sync();
    }
}

// ---- LDS-tiled path ---------------------------------------------------------------------------------------

// a tile the fast path cannot take (its reference span straddles the origin of a circular chromosome): queued for
// k_materialise_slow
struct SlowTile { uint32_t read, piece, m0, m1; };
struct SlowQueue { SlowTile *items; uint32_t *count; uint32_t cap; };

#define NS_REF_PAD 64u       // bytes allocated before and after the reference bases: any 16-byte load that overlaps a segment is in bounds

__device__ __forceinline__ uint32_t bfi(uint32_t mask, uint32_t a, uint32_t b) {     // (mask & a) | (~mask & b), one instruction
    uint32_t d;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "v"(mask), "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ uint32_t and_or(uint32_t a, uint32_t mask, uint32_t c) { return (a & mask) | c; }    // v_and_or_b32

// what a piece is copied from
enum { MAT_REF = 0,          // the reference -> the record (qualities drawn here if FASTQ)
       MAT_HP_SCRATCH = 1,   // -k, first pass: the reference -> the pre-homopolymer read in the scratch buffer (class bits kept, no qualities)
       MAT_HP_FINAL = 2 };   // -k, second pass: the scratch read + the homopolymer edits as its event list -> the record

// ---- qualities (predict_base_qualities, bq:120-130, the truncated log-normal of bq:9-20; classes S:1421-1423, 1953-1955) --------------------------------------------------
// The 16-bit draw of emitted piece position m is halfword m & 7 of Philox(ST_QUAL, sid, attempt, idx = m >> 3).
__device__ __forceinline__ uint32_t dpp_wave_shr1(uint32_t lane0_value, uint32_t v) {      // lane l gets v of lane l - 1, lane 0 gets lane0_value
    return (uint32_t)__builtin_amdgcn_update_dpp((int)lane0_value, (int)v, 0x138, 0xf, 0xf, false);
}
typedef uint16_t ns_v2u16 __attribute__((ext_vector_type(2)));
struct QualState {
    const uint16_t *lut;                 // LDS: NS_QLUT_SLOTS x 1024 bucket entries (see qual_value_lut), slot = class
};

// 16 qualities from the 16 draws D and the class slot of every byte (cs[k]: slot << 3 in each byte of dword k of the chunk)
__device__ __forceinline__ void qual_lookup16(const QualState &Q, const DevModel &m, const uint32_t D[8], const uint32_t cs[4], bool skip_lut,
                                              uint64_t &qlo, uint64_t &qhi) {
    const uint8_t *lut = reinterpret_cast<const uint8_t *>(Q.lut);
    uint32_t Q2[8], flags = 0;
#pragma unroll
    for (uint32_t k = 0; k < 8; ++k) {
        const uint32_t d = D[k], cw = cs[k >> 1];
        // (slot << 3) of byte 2k / 2k + 1 moved to bits 11.. = slot * 2048 bytes
        const uint32_t o0 = __builtin_amdgcn_perm(0u, cw, (k & 1u) ? 0x0c0c020cu : 0x0c0c000cu);
        const uint32_t o1 = __builtin_amdgcn_perm(0u, cw, (k & 1u) ? 0x0c0c030cu : 0x0c0c010cu);
        const uint32_t a0 = and_or(d >> 5, 0x7feu, o0), a1 = and_or(d >> 21, 0x7feu, o1);        // byte address of entry [slot][h >> 6]
        uint32_t E;
        if (skip_lut) E = (a0 & 0x3f80u) | (a1 & 0x3f80u) << 16 | 0x00400040u;
        else {                               // (built as a two-halfword vector: one v_perm_b32 instead of shift + or)
            ns_v2u16 e2;
            e2.x = *reinterpret_cast<const uint16_t *>(lut + a0); e2.y = *reinterpret_cast<const uint16_t *>(lut + a1);
            E = __builtin_bit_cast(uint32_t, e2);
        }
        flags |= E;
        // see qual_value_lut — on both halfwords at once (packed 16-bit add and shift: no carry between the halves, nothing to mask; an
        // entry with bit 15 set gives garbage here and is redone by the exact walk below)
        const ns_v2u16 s2 = (__builtin_bit_cast(ns_v2u16, E) + __builtin_bit_cast(ns_v2u16, d & 0x003f003fu)) >> (uint16_t)7;
        Q2[k] = __builtin_bit_cast(uint32_t, s2);
    }
    if (__ballot((flags & 0x80008000u) != 0)) {                        // a bucket with several thresholds (never with the loader's tables)
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) {
            const uint32_t c0 = (cs[k >> 1] >> (16 * (k & 1) + 3)) & 7u, c1 = (cs[k >> 1] >> (16 * (k & 1) + 11)) & 7u;      // slot = class
            const uint32_t q0 = qual_value(m.qual_thr + min(c0, (uint32_t)NS_Q_COUNT - 1u) * NS_QUAL_LEVELS, D[k] & 0xffffu);
            const uint32_t q1 = qual_value(m.qual_thr + min(c1, (uint32_t)NS_Q_COUNT - 1u) * NS_QUAL_LEVELS, D[k] >> 16);
            Q2[k] = q0 | q1 << 16;
        }
    }
    const uint32_t b0 = __builtin_amdgcn_perm(Q2[1], Q2[0], 0x06040200u), b1 = __builtin_amdgcn_perm(Q2[3], Q2[2], 0x06040200u);
    const uint32_t b2 = __builtin_amdgcn_perm(Q2[5], Q2[4], 0x06040200u), b3 = __builtin_amdgcn_perm(Q2[7], Q2[6], 0x06040200u);
    qlo = (uint64_t)b0 | (uint64_t)b1 << 32; qhi = (uint64_t)b2 | (uint64_t)b3 << 32;
}
// ---- the class stream between the record kernel and k_qualities --------------------------------------------------------------------
// The record kernel of a FASTQ batch writes the bases; the class of every base (match / substituted / inserted: which quality model
// applies, S:1953-1955) is on the bytes of its LDS tile as two bits (NS_CLS_MIS_BIT, NS_CLS_INS_BIT) and leaves as ONE 32-bit word per
// 16-byte chunk: byte i of the chunk in bits 2i (substituted) and 2i + 1 (inserted).  k_qualities reads the words back and draws the
// quality line.  (As ONE kernel the quality draws — two Philox blocks in flight per lane, sixteen table reads — held the record kernel
// at 124 VGPRs = 4 wavefronts per SIMD, where it issued 52 % of the time: FASTQ 14.2 ms per 950 000 reads against 5.4 for FASTA.)
__device__ __forceinline__ uint32_t cls_pack16(uint32_t r0, uint32_t r1, uint32_t r2, uint32_t r3) {
    auto field = [](uint32_t r) {            // the four 2-bit fields of a dword gathered into its top byte (no two terms of the product meet)
        const uint32_t t = and_or(r >> 4, 0x02020202u, (r >> 3) & 0x01010101u);
        return t * 0x01041040u;
    };
    const uint32_t p0 = field(r0), p1 = field(r1), p2 = field(r2), p3 = field(r3);
    return __builtin_amdgcn_perm(p1, p0, 0x0c0c0703u) | __builtin_amdgcn_perm(p3, p2, 0x07030c0cu);
}
// dword k of the chunk: the table slot of every byte as slot << 3 (the form qual_lookup16 takes)
__device__ __forceinline__ uint32_t cls_unpack4(uint32_t w, uint32_t k) {
    const uint32_t e = (w >> (8u * k)) & 0xffu;
    uint32_t t = e << 3;
    t |= e << 9; t |= e << 15; t |= e << 21;
    return t & 0x18181818u;
}
// class words of read r: a piece of n bytes has at most n / 16 + 2 chunks (its first and last may be partial) and a FASTQ record is
// longer than twice its bases, so the words of a read fit behind (its record offset) / 16 once every read brings two words per piece it
// can have (per_read = 2, chimeric batches 2 * (2 * NS_MAX_SEG - 1)); piece pi starts at output offset q of the read.  (Not the read's
// piece_off: a chimeric read that was planned again lies behind the others, out of read order.)
__device__ __forceinline__ uint64_t cls_word0(uint64_t rec_off, uint64_t r, uint32_t per_read) { return (rec_off >> 4) + r * per_read; }
__device__ __forceinline__ uint32_t cls_per_read(const ns_params &prm) { return prm.chimeric ? 2u * (2u * NS_MAX_SEG - 1u) : 2u; }

// the bucket tables of the classes a piece can hold, global -> LDS (all threads of the workgroup; the caller synchronises)
__device__ __forceinline__ void qual_lut_load(uint16_t *lds, const DevModel &m, uint32_t tid, uint32_t nthreads) {
    const uint4 *src = reinterpret_cast<const uint4 *>(m.qual_lut);
    uint4 *dst = reinterpret_cast<uint4 *>(lds);
    for (uint32_t i = tid; i < NS_QLUT_SLOTS * 128u; i += nthreads) dst[i] = src[i];      // slot = class (NS_Q_*)
}

__device__ __forceinline__ void lds_or32(uint8_t *p, uint32_t v) {
    __hip_atomic_fetch_or(reinterpret_cast<unsigned int *>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}
__device__ __forceinline__ void lds_or64(uint8_t *p, uint64_t v) {
    __hip_atomic_fetch_or(reinterpret_cast<unsigned long long *>(p), (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

// letter word of event j of the piece: word(j) = Philox(ST_SUB, seg, attempt, idx = j >> 2).w[j & 3].  The record
// kernel draws it where it stages the events (one block per lane and tile: a k_words pass that wrote the words for the next kernel to
// read back cost 0.65 ms and 1.4 KB of traffic per read); the second pass of -k takes the words k_hp_drain filed with its edits.
template <int MODE>
__device__ __forceinline__ uint32_t event_word(const PieceCtx &pc, const ns_key &key, uint32_t a, uint32_t j) {
    if constexpr (MODE == MAT_HP_FINAL) return pc.wd[j];
    else { const u32x4 w = ns_draw(key, ST_SUB, pc.sid, a, j >> 2, 0); return ns_word(w, j & 3u); }
}
// ================================================================================================================================
// Version 7 of the tile (round 6).  The tile's copy is cut into SUB-RUNS — maximal stretches of output bytes that lie in one 16-byte chunk
// AND under one event; a sub-run is ONE unaligned 16-byte load at the chunk's source position under the event's shift, masked to its bytes.
// Version 6 (rounds 3-5) made every sub-run an element of a sorted list (chunk starts + event starts, <= 191 = three passes of the
// wavefront per tile), loaded by the lane of its ELEMENT, OR-ed into the LDS tile and read back by the lane of its CHUNK.  Here the chunk
// lane loads the sub-run that is in force at the chunk's first byte ITSELF and keeps it in registers — for a chunk no event starts in
// (more than half of the chunks of an aligned read, 19 of 20 in the second pass of -k) that is the whole chunk: one load, complement /
// reverse, one aligned store, no LDS round trip for the data.  Only the sub-runs that start INSIDE a chunk (one per event, lane = event,
// one pass) and the letters go through the LDS tile; the chunk lane ORs its own bytes with what it finds there.  No element list, no
// deferred final pass, 3.7 instead of 5.0 KB of LDS per wavefront.  Same-box A/B (profiles/r06/ab_tile_v7.log): record kernel 5.23 ->
// 5.11 ms, bytes identical; other tile sizes / occupancies of v7 all lose (one chunk per lane at 8 waves 6.5 ms, four at 6 waves 6.8 ms,
// two at 8 waves with 64 VGPRs — spills — 6.0 ms).
// ================================================================================================================================
template <uint32_t TC>
struct __align__(16) TileLds7 {
    static constexpr uint32_t T_OUT_ = 1024u * TC;          // output bytes of a tile: TC 16-byte chunks per lane
    uint32_t mlut[17][4];                       // mlut[i]: 16-byte mask with bytes >= i set; [16] empty
    uint2 ent[T_EV + 1];                        // per staged event (0: the event in force at the tile start): x = first output offset copied
                                                // under it, y = segment position minus output offset of those bytes
    uint32_t eos[T_EV + 1];                     // first output offset of tile event k (0xffffffff behind the last)
    uint32_t hist[64 * TC];         // build: last event at or before chunk c (+1); afterwards: the number of events at or before it
    __align__(16) uint8_t out[1024u * TC + 16 + 64]; // letters and event sub-runs of the tile (zero where nothing has been written) + dump slots
};
template <uint32_t TC>
__device__ __forceinline__ void tile_lds_init(TileLds7<TC> &T, uint32_t lane) {
    for (uint32_t c = lane * 16; c < 1024u * TC + 16 + 64; c += 64 * 16) *reinterpret_cast<uint4 *>(&T.out[c]) = make_uint4(0, 0, 0, 0);
    if (lane < 17) {
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k)
            T.mlut[lane][k] = lane <= 4 * k ? 0xffffffffu : lane >= 4 * k + 4 ? 0u : 0xffffffffu << (8 * (lane - 4 * k));
    }
}
// IUPAC codes among the bytes [i0, i1) of a masked 16-byte group whose byte b lies at segment position x0 + b (case_convert, S:743-755: rare)
__device__ __forceinline__ void resolve16(uint32_t &a0, uint32_t &a1, uint32_t &a2, uint32_t &a3, uint32_t i0, uint32_t i1, uint32_t x0,
                                          const ns_key &key, uint32_t sid, uint32_t a) {
    for (uint32_t b = i0; b < i1; ++b) {
        uint32_t wk = b < 8 ? (b < 4 ? a0 : a1) : (b < 12 ? a2 : a3);
        const uint32_t ch = (wk >> (8 * (b & 3))) & 0xff;
        if (!(ch & 0x80u)) continue;
        const uint32_t r = resolve_base(ch, key, sid, a, x0 + b);
        wk = (wk & ~(0xffu << (8 * (b & 3)))) | r << (8 * (b & 3));
        if (b < 4) a0 = wk; else if (b < 8) a1 = wk; else if (b < 12) a2 = wk; else a3 = wk;
    }
}
// Phase-ablation bits of the record kernel (NS_DEBUG_SKIP: 1 no chunk loads / final step, 2 no letters, 16 no stores, 32 no event sub-run merge,
// 64 no letter-word Philox, 128 no event sub-run loads, 256 no chunk-lane look-ups and loads): compiled in only with -DNS_ABLATE (scripts/ablate_materialise.sh builds that variant) — every test of a
// run-time flag in the tile loop is a branch and a live SGPR in a kernel that spills ~80 of them.
#ifdef NS_ABLATE
#define NS_DBG(bit) (dbg & (bit))
#else
#define NS_DBG(bit) false
#endif
// a & b & ~c in one instruction (v_bitop3_b32, truth table 0x40 for inputs 0xF0, 0xCC, 0xAA; the compiler emits not + and + and)
__device__ __forceinline__ uint32_t and_andn(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x40); }
template <bool FASTQ, int MODE, uint32_t TC>
__device__ inline void materialise_piece7(const DevModel &m, const DevRef &ref, TileLds7<TC> &T, const ReadOut &ro, const ns_key &key,
                                          uint32_t a, const PieceCtx &pc, uint32_t pq, uint32_t lane, uint32_t dbg, const SlowQueue &sq,
                                          uint32_t read_idx, uint32_t piece_idx, uint32_t *__restrict__ cls) {
    constexpr bool CLSOUT = FASTQ && MODE != MAT_HP_SCRATCH;           // the class of every base leaves as 2 bits for k_qualities (cls: the piece's words)
    constexpr bool HPF = MODE == MAT_HP_FINAL;
    constexpr uint32_t T_OUT_ = 1024u * TC;
    uint32_t jb = 0;                       // events with out_start < M0
    uint32_t L0_out = 0, L0_rp = 0, L0_pt = 3u << 12, L0_wd = 0, L0_j = 0;   // the event in force at M0 (synthetic start: no payload, copy from 0)
    const uint8_t *seg0 = ref.bases + pc.chrom_base + pc.pos;           // segment position 0
    const bool wraps = !HPF && pc.pos + pc.ref_len > pc.chrom_len;
    const uint32_t wrap_at = wraps ? (uint32_t)(pc.chrom_len - pc.pos) : 0xffffffffu;   // first segment position beyond the origin
    // output offsets m with m = phi (mod 16) start an aligned 16-byte group of the destination
    const uint32_t phi = ro.reversed ? ((uint32_t)(uintptr_t)ro.seq + ro.seq_len - pq) & 15u : (0u - ((uint32_t)(uintptr_t)ro.seq + pq)) & 15u;
    ns_event e_pre; e_pre.pos = 0; e_pre.info = 0; uint32_t w_pre = 0;
    if (lane < pc.n_ev) e_pre = pc.ev[lane];
    auto cached_word = [&](uint32_t j0) -> uint32_t { if (NS_DBG(64u)) return 0u; const uint32_t j = j0 + lane; return event_word<MODE>(pc, key, a, j < pc.n_ev ? j : 0u); };
    w_pre = cached_word(0u);
    uint32_t cls_carry = 0;                // class bits of the chunk the last tile ended in (tiles queued for the generic path: none)
    for (uint32_t M0 = 0; M0 < pc.out_len;) {
        const uint32_t A0 = M0 - ((M0 - phi) & 15u);             // aligned origin of the tile (<= M0; may be "negative" = wrapped)
        uint32_t M1 = min(A0 + T_OUT_, pc.out_len);
        // ---- 1. events of the tile
        const ns_event e = e_pre;
        const uint32_t e_wd = w_pre;
        const bool valid = jb + lane < pc.n_ev;
        const uint32_t os = ev_out_start(e), len = ns_ev_len(e.info), ty = ns_ev_type(e.info);
        const uint32_t e_pt = (ty == NS_DEL ? 0u : len) | ty << 12, e_rp = e.pos + (ty == NS_INS ? 0u : len);
        if (jb + 63 < pc.n_ev) {           // more events than lanes: the tile ends early — on a CHUNK boundary, so that the next tile starts on one
            const uint32_t os63 = (uint32_t)__builtin_amdgcn_readlane((int)os, 63);
            if (os63 < M1) {
                const uint32_t cut = A0 + ((os63 - A0) & ~15u);
                M1 = (int32_t)(cut - M0) > 0 ? cut : os63;                                // (A0, and with it cut, may be "negative": wrapped)
            }
        }
        const bool take = valid && os < M1;
        const uint32_t cnt = (uint32_t)__popcll(__ballot(take));
        if (M1 <= M0) {                    // 64 events at one output offset (zero-length matches between deletions): not a case
            cls_carry = 0;                 // for the tile machinery; the generic path takes the tile
            M1 = min(M0 + T_OUT_, pc.out_len);
            uint32_t j2 = jb;
            while (j2 < pc.n_ev && ev_out_start(pc.ev[j2]) < M1) ++j2;
            if (lane == 0) {
                const uint32_t slot = atomicAdd(sq.count, 1u);
                if (slot < sq.cap) sq.items[slot] = SlowTile{read_idx, piece_idx, M0, M1};
            }
            if (j2 > jb) {
                const ns_event le = pc.ev[j2 - 1];
                const uint32_t ll = ns_ev_len(le.info), lt = ns_ev_type(le.info);
                L0_out = uni(ev_out_start(le)); L0_pt = uni((lt == NS_DEL ? 0u : ll) | lt << 12); L0_rp = uni(le.pos + (lt == NS_INS ? 0u : ll));
                L0_wd = uni(event_word<MODE>(pc, key, a, j2 - 1)); L0_j = uni(j2 - 1);
            }
            jb = uni(j2); M0 = M1;
            e_pre.pos = 0; e_pre.info = 0; w_pre = 0;
            if (jb + lane < pc.n_ev) e_pre = pc.ev[jb + lane];
            w_pre = cached_word(jb);
            continue;
        }
        const uint32_t nc = (M1 - A0 + 15u) >> 4;                 // chunks of the tile (chunk 0 starts at M0, chunk c > 0 at A0 + 16 c)
        // chunk an event sorts in front of: the first one that starts at or after it (an event AT a chunk start comes first, so the
        // chunk already copies under it)
        const uint32_t ekey = take ? (os <= M0 ? 0u : (os - A0 + 15u) >> 4) : 0xffffffffu;
        const uint32_t s1 = os + (e_pt & 0xfffu), y1 = e_rp - s1;
        {
            const uint32_t s0 = L0_out + (L0_pt & 0xfffu);
            T.ent[0] = make_uint2(s0, L0_rp - s0);                // (every lane, same value)
            if (take) T.ent[1 + lane] = make_uint2(s1, y1);
            T.eos[lane] = take ? os : 0xffffffffu;                // (cnt <= 63: slot cnt holds the end mark)
        }
#pragma unroll
        for (uint32_t t = 0; t < TC; ++t) T.hist[64 * t + lane] = 0;
        const uint32_t jb_next = jb + cnt;
        e_pre.pos = 0; e_pre.info = 0; w_pre = 0;
        if (M1 < pc.out_len && jb_next + lane < pc.n_ev) e_pre = pc.ev[jb_next + lane];       // prefetch for the next tile
        if (M1 < pc.out_len) w_pre = cached_word(jb_next);
        wave_sync();
        // first output offset of the next event of the tile (the last one: none)
        const uint32_t os_next = dpp_wave_shl1(0xffffffffu, take ? os : 0xffffffffu);
        {   // hist[c] = number of the tile's events sorted in front of chunk c — written by the LAST one (sorted events: lane l is event l + 1
            // of the tile), the chunks in between inherit it through a prefix maximum
            const uint32_t c_next = dpp_wave_shl1(0xffffffffu, ekey);
            if (ekey < 64 * TC && ekey != c_next) T.hist[ekey] = lane + 1u;
        }
        // ---- the event in force at M1 (wave-uniform): the last one taken, straight from its lane's registers
        uint32_t osl = L0_out, ptl = L0_pt, rpl = L0_rp, wdl = L0_wd, jl = L0_j;
        if (cnt) {
            osl = (uint32_t)__builtin_amdgcn_readlane((int)os, (int)(cnt - 1));
            ptl = (uint32_t)__builtin_amdgcn_readlane((int)e_pt, (int)(cnt - 1));
            rpl = (uint32_t)__builtin_amdgcn_readlane((int)e_rp, (int)(cnt - 1));
            wdl = (uint32_t)__builtin_amdgcn_readlane((int)e_wd, (int)(cnt - 1));
            jl = jb + cnt - 1;
        }
        const uint8_t *tb = seg0 - 32;                             // + 32 in the lane offsets: they never go negative
        bool fast = true;
        if (wraps) {                                               // reference span of the tile only matters next to the origin
            const uint32_t pl0 = L0_pt & 0xfffu, ty0 = L0_pt >> 12, d0 = M0 - L0_out;
            const uint32_t x0 = (d0 < pl0 && ty0 == NS_INS) ? L0_rp : L0_rp + d0 - pl0;
            const uint32_t pll = ptl & 0xfffu, dl = M1 - osl;
            uint32_t x1 = dl <= pll ? rpl : rpl + (dl - pll);
            if (x1 < x0) x1 = x0;
            if (x0 >= wrap_at) tb -= pc.chrom_len;                 // whole tile beyond the origin
            else if (x1 > wrap_at) fast = false;                   // tile straddles the origin
        }
        if (!fast) {
            cls_carry = 0;
            if (lane == 0) {
                const uint32_t slot = atomicAdd(sq.count, 1u);
                if (slot < sq.cap) sq.items[slot] = SlowTile{read_idx, piece_idx, M0, M1};
            }
            L0_out = osl; L0_rp = rpl; L0_pt = ptl; L0_wd = wdl; L0_j = jl;
            jb = jb_next; M0 = M1;
            wave_sync();
            continue;
        }
        wave_sync();
        // ---- 3a. lane per chunk: the sub-run in force at the chunk's first byte — [max(chunk start, first byte copied under the last event at
        // or before it), min(next event, chunk end)) — is loaded HERE and stays in registers until the chunk leaves
        uint4 f[TC]; uint32_t fi[TC], fy[TC];
        {
            uint32_t scan_base = 0;
#pragma unroll
            for (uint32_t t = 0; t < TC; ++t) {
                const uint32_t ci = 64 * t + lane;
                const uint32_t incl = max(scan_base, wave_incl_max(T.hist[ci]));
                scan_base = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                f[t] = make_uint4(0, 0, 0, 0); fi[t] = 16u | 16u << 8; fy[t] = 0;
                if (ci < nc && !NS_DBG(1u) && !NS_DBG(256u)) {
                    const uint32_t cs = A0 + 16u * ci, p = ci == 0 ? M0 : cs;
                    const uint2 E = T.ent[incl];
                    const uint32_t nx = T.eos[incl];
                    const uint32_t st = max(p, E.x), en = min(nx, min(cs + 16u, M1));
                    if ((int32_t)(en - st) > 0) {
                        fi[t] = (st - cs) | (en - cs) << 8; fy[t] = E.y;
                        __builtin_memcpy(&f[t], tb + (E.y + cs + 32u), 16);
                    }
                }
            }
        }
        // ---- 3b. lane per event: the sub-run that starts behind the event's letters, up to the next event or the end of its chunk — unless the
        // chunk starts under this event (then the chunk lane has it)
        uint4 fe = make_uint4(0, 0, 0, 0); uint32_t ei0 = 16, ei1 = 16, eoc = 0;
        if (take && !NS_DBG(1u) && !NS_DBG(128u)) {
            const uint32_t c = (s1 - A0) >> 4, ecs = A0 + 16u * c;
            const uint32_t en = min(os_next, min(ecs + 16u, M1));
            if ((int32_t)(en - s1) > 0 && ekey > c) {
                eoc = 16u * c; ei0 = s1 - ecs; ei1 = en - ecs;
                __builtin_memcpy(&fe, tb + (y1 + ecs + 32u), 16);
            }
        }
        // ---- 2. letters: lane per event; lane 63 (never a taker) continues the payload of the event in force at M0
        if (!NS_DBG(2u)) {
            const bool cont = lane == 63 && L0_out + (L0_pt & 0xfffu) > M0;
            const uint32_t b_os = cont ? L0_out : os, b_pt = cont ? L0_pt : e_pt, b_rp = cont ? L0_rp : e_rp;
            const uint32_t b_j = cont ? L0_j : jb + lane;
            uint32_t frac = cont ? L0_wd : e_wd;
            const uint32_t b_pl = b_pt & 0xfffu, b_ty = b_pt >> 12;
            const bool on = (take || cont) && b_pl;
            const uint32_t xs = b_rp - b_pl;                      // segment position under the first substituted base
            // fast path (branch-free): up to EIGHT letters, all inside the tile, plain bases under a substitution
            const bool mis = b_ty == NS_MIS;
            bool fast_l = on && b_pl <= 8 && b_os >= M0 && b_os + b_pl <= M1 && !wraps;
            uint2 cur8 = make_uint2(0x41414141u, 0x41414141u);
            if (fast_l && mis) __builtin_memcpy(&cur8, seg0 + xs, 8);
            if constexpr (!HPF) fast_l = fast_l && !((cur8.x | cur8.y) & 0x80808080u);
            if (fast_l) {
                uint32_t f3 = frac, L2[2];
#pragma unroll
                for (uint32_t g = 0; g < 2; ++g) {
                    const uint32_t cur4 = g ? cur8.y : cur8.x;
                    const uint32_t x8 = (frac >> (8u * g)) & 0xffu, t8 = (x8 | x8 << 12) & 0x000f000fu;
                    const uint32_t ins4 = __builtin_amdgcn_perm(0u, 0x47435441u, (t8 | t8 << 6) & 0x03030303u);          // S:1990
                    const uint32_t d0 = next_digit3(f3), d1 = next_digit3(f3), d2 = next_digit3(f3), d3 = next_digit3(f3);
                    const uint32_t d4 = d0 | d1 << 8 | d2 << 16 | d3 << 24;
                    const uint32_t vv = (cur4 >> 1) & 0x03030303u;                           // A 0, C 1, T 2, G 3
                    const uint32_t rank4 = (vv & 0x01010101u) << 1 | ((vv >> 1) & 0x01010101u);   // rank in "ATCG": A 0, T 1, C 2, G 3
                    const uint32_t ge = ((d4 | 0x80808080u) - rank4) & 0x80808080u;          // per byte: digit >= rank
                    const uint32_t mis4 = __builtin_amdgcn_perm(0u, 0x47435441u, d4 + (ge >> 7));                        // S:1968-1972
                    uint32_t letters = mis ? mis4 : ins4;
                    if constexpr (FASTQ) {                             // the quality class travels with the letter
                        uint32_t cls4 = mis ? 0x01010101u * NS_CLS_MIS_BIT : 0x01010101u * NS_CLS_INS_BIT;
                        if constexpr (HPF) {
                            if (mis) cls4 = (frac & 1u) ? NS_CLS_MIS_BIT : (cur4 & (NS_CLS_MIS_BIT | NS_CLS_INS_BIT));
                            else if (g == 0 && (frac >> 31)) cls4 = (cls4 & ~0xffu) | NS_CLS_MIS_BIT;
                        }
                        letters |= cls4;
                    }
                    L2[g] = letters;
                }
                // the <= 8 letters leave as three aligned dword ORs into the (zeroed) tile instead of eight predicated byte stores: no other
                // sub-run or letter shares a byte with them, so the OR is a store
                const uint64_t keep = b_pl >= 8u ? ~0ull : ~(~0ull << (8u * b_pl));
                const uint64_t v = ((uint64_t)L2[0] | (uint64_t)L2[1] << 32) & keep;
                const uint32_t o = b_os - A0, sh = 8u * (o & 3u);
                const uint64_t lo64 = v << sh;
                const uint32_t top = (uint32_t)(((v >> 32) << sh) >> 32);
                uint8_t *w = &T.out[o & ~3u];
                lds_or32(w, (uint32_t)lo64); lds_or32(w + 4, (uint32_t)(lo64 >> 32)); lds_or32(w + 8, top);
            }
            if (on && !fast_l) {                                  // long payloads, tile borders, IUPAC under a substitution, the origin
                const uint32_t i_lo = b_os < M0 ? M0 - b_os : 0u;
                const uint32_t i_hi = min(b_pl, M1 - b_os);
                const uint32_t word0 = frac;
                for (uint32_t i = 0; i < i_hi; ++i) {
                    if (i && !(i & 15)) frac = payload_word(key, pc.sid, a, b_j, i >> 4);
                    uint32_t b;
                    if (b_ty == NS_INS) {
                        b = bases_atcg((frac >> (2 * (i & 15))) & 3u);
                        if constexpr (FASTQ) b |= (HPF && i == 0 && (word0 >> 31)) ? NS_CLS_MIS_BIT : NS_CLS_INS_BIT;
                    } else {
                        const uint32_t x = xs + i;
                        const uint32_t src = ref_base_at(ref, pc, x);
                        if constexpr (HPF) {
                            b = mis_from_digit(src & ~(NS_CLS_MIS_BIT | NS_CLS_INS_BIT), next_digit3(frac));
                            if constexpr (FASTQ) b |= (word0 & 1u) ? NS_CLS_MIS_BIT : (src & (NS_CLS_MIS_BIT | NS_CLS_INS_BIT));
                        } else {
                            b = mis_from_digit(resolve_base(src, key, pc.sid, a, x), next_digit3(frac));
                            if constexpr (FASTQ) b |= NS_CLS_MIS_BIT;
                        }
                    }
                    if (i >= i_lo) T.out[b_os + i - A0] = (uint8_t)b;
                }
            }
        }
        // ---- 3c. the event sub-runs arrive: masked to their bytes, IUPAC codes resolved, OR-ed into the tile (no two sub-runs share a byte:
        // the OR is a store that needs no ordering)
        if (cnt && ei0 < 16u && !NS_DBG(32u)) {
            const uint4 m0 = *reinterpret_cast<const uint4 *>(&T.mlut[ei0][0]), m1 = *reinterpret_cast<const uint4 *>(&T.mlut[ei1][0]);
            uint32_t a0 = and_andn(fe.x, m0.x, m1.x), a1 = and_andn(fe.y, m0.y, m1.y), a2 = and_andn(fe.z, m0.z, m1.z), a3 = and_andn(fe.w, m0.w, m1.w);
            if constexpr (!HPF) {
                if ((a0 | a1 | a2 | a3) & 0x80808080u) resolve16(a0, a1, a2, a3, ei0, ei1, A0 + eoc + y1, key, pc.sid, a);
            }
            lds_or64(&T.out[eoc], (uint64_t)a0 | (uint64_t)a1 << 32);
            lds_or64(&T.out[eoc + 8], (uint64_t)a2 | (uint64_t)a3 << 32);
        }
        wave_sync();
        // ---- 4. lane per aligned 16-byte chunk: own sub-run | what the tile holds; qualities classes, complement / reverse, one aligned store
        // (unrolled: f[t] must stay in registers)
#pragma unroll
        for (uint32_t t = 0; t < TC; ++t) {
            const uint32_t ci = 64 * t + lane;
            const uint32_t c0 = A0 + 16 * ci;                         // chunk origin (chunk 0 of a piece's first tile may start before M0)
            const uint32_t lo_m = ci == 0 ? M0 : c0, hi_m = min(c0 + 16, M1);
            const bool active = (int32_t)(hi_m - lo_m) > 0 && !NS_DBG(1u);
            uint32_t cw = 0;
            if (active) {
                const uint4 v = *reinterpret_cast<const uint4 *>(&T.out[16 * ci]);
                *reinterpret_cast<uint4 *>(&T.out[16 * ci]) = make_uint4(0, 0, 0, 0);      // the tile is left clean for the next one
                const uint32_t i0 = fi[t] & 0xffu, i1 = fi[t] >> 8;
                const uint4 m0 = *reinterpret_cast<const uint4 *>(&T.mlut[i0][0]), m1 = *reinterpret_cast<const uint4 *>(&T.mlut[i1][0]);
                uint32_t a0 = and_andn(f[t].x, m0.x, m1.x), a1 = and_andn(f[t].y, m0.y, m1.y), a2 = and_andn(f[t].z, m0.z, m1.z), a3 = and_andn(f[t].w, m0.w, m1.w);
                if constexpr (!HPF) {
                    if ((a0 | a1 | a2 | a3) & 0x80808080u) resolve16(a0, a1, a2, a3, i0, i1, c0 + fy[t], key, pc.sid, a);
                }
                if (NS_DBG(512u)) { a0 = a1 = a2 = a3 = 0x41414141u; }
                uint32_t r0 = v.x | a0, r1 = v.y | a1, r2 = v.z | a2, r3 = v.w | a3;
                uint64_t qlo = 0, qhi = 0;
                uint32_t s0 = lo_m - c0, count = hi_m - lo_m;          // bytes [s0, s0 + count) of the chunk are this tile's
                if constexpr (CLSOUT) {                                // chunk k of the piece covers its positions [16 k - g, 16 k - g + 16), g = -phi mod 16
                    cw = cls_pack16(r0, r1, r2, r3);
                    if (ci == 0) cw |= cls_carry;                      // (a tile cut inside a chunk: the bits of the part the last tile wrote)
                    if (!pc.kind) cls[(c0 + ((0u - phi) & 15u)) >> 4] = cw;
                    r0 &= NS_CLS_STRIP; r1 &= NS_CLS_STRIP; r2 &= NS_CLS_STRIP; r3 &= NS_CLS_STRIP;
                }
                uint64_t lo = (uint64_t)r0 | (uint64_t)r1 << 32, hi = (uint64_t)r2 | (uint64_t)r3 << 32;
                if (s0) {                                              // front-partial chunk (first chunk of a piece): shift down
                    const uint32_t sh = 8 * s0;
                    if (sh < 64) { lo = (lo >> sh) | (hi << (64 - sh)); hi >>= sh; }
                    else { lo = hi >> (sh - 64); hi = 0; }
                }
                if (!NS_DBG(16u)) { PendingChunk pd = prep_chunk(ro, pq + lo_m, count, lo, hi, qlo, qhi); flush_chunk(ro, pd); }
            }
            if constexpr (CLSOUT) {                                    // the tile ends inside a chunk: its word goes on in the next tile's chunk 0
                const uint32_t cl = (M1 - 1u - A0) >> 4;
                if (t == (cl >> 6)) cls_carry = (M1 < pc.out_len && ((M1 - A0) & 15u)) ? (uint32_t)__builtin_amdgcn_readlane((int)cw, (int)(cl & 63u)) : 0u;
            }
        }
        L0_out = osl; L0_rp = rpl; L0_pt = ptl; L0_wd = wdl; L0_j = jl;
        jb = jb_next; M0 = M1;
        wave_sync();
    }
}

// ---- the quality lines (k_qualities) ----------------------------------------------------------------------------------------------
// One lane per 16 positions of a quality stream, [16 j, 16 j + 16): exactly its Philox blocks 2 j and 2 j + 1 (the 16-bit draw of position
// m is halfword m & 7 of block m >> 3), sixteen bucket look-ups in LDS, one 16-byte store (the quality line starts wherever the sequence
// line ends: no grid is aligned with it).
__device__ __forceinline__ void quals16(const QualState &Q, const DevModel &m, const ns_key &key, uint32_t stream, uint32_t sid, uint32_t a,
                                        uint32_t j, const uint32_t cs[4], uint64_t &qlo, uint64_t &qhi) {
    const u32x4 k1 = ns_draw(key, stream, sid, a, 2u * j, 0), k2 = ns_draw(key, stream, sid, a, 2u * j + 1u, 0);
    const uint32_t D[8] = {k1.x, k1.y, k1.z, k1.w, k2.x, k2.y, k2.z, k2.w};
    qual_lookup16(Q, m, D, cs, false, qlo, qhi);
}
// A piece: the class words lie on the record kernel's chunk grid, g positions ahead of the piece's positions — two neighbouring words,
// funnel-shifted; the next iteration's words are in flight during this one's draws.  kind != 0 (the gap of a chimeric read): every base
// is of class `unmapped` (S:1564), no class words.
__device__ inline void qualities_piece(const DevModel &m, const QualState &Q, const ReadOut &ro, const ns_key &key, uint32_t a, uint32_t sid,
                                       uint32_t kind, uint32_t out_len, uint32_t pq, const uint32_t *__restrict__ cls, uint32_t lane) {
    const uint32_t phi = ro.reversed ? ((uint32_t)(uintptr_t)ro.seq + ro.seq_len - pq) & 15u : (0u - ((uint32_t)(uintptr_t)ro.seq + pq)) & 15u;
    const uint32_t g2 = 2u * ((0u - phi) & 15u);               // word k of the piece: positions [16 k - g, 16 k - g + 16)
    const uint32_t nchunks = (out_len + 15u) >> 4;
    uint32_t wl = 0, wh = 0;
    if (!kind && lane < nchunks) { wl = cls[lane]; wh = cls[lane + 1u]; }      // (the word behind the piece's last: allocated, any value)
    for (uint32_t j = lane; j < nchunks; j += 64) {
        const uint32_t w = __builtin_amdgcn_alignbit(wh, wl, g2);
        if (!kind && j + 64u < nchunks) { wl = cls[j + 64u]; wh = cls[j + 65u]; }
        uint32_t cs[4];
        if (kind) cs[0] = cs[1] = cs[2] = cs[3] = 0x08080808u * (uint32_t)NS_Q_UNMAPPED;
        else { cs[0] = cls_unpack4(w, 0); cs[1] = cls_unpack4(w, 1); cs[2] = cls_unpack4(w, 2); cs[3] = cls_unpack4(w, 3); }
        uint64_t qlo, qhi;
        quals16(Q, m, key, ST_QUAL, sid, a, j, cs, qlo, qhi);
        store_qual_chunk(ro, pq + 16u * j, min(16u, out_len - 16u * j), qlo, qhi);
    }
}
// head and tail (S:1421-1423): ONE stream of head + tail draws of class `ht`; the first `head` go to the start of the read, the rest to its
// end — the chunk that holds both is stored in two parts
__device__ inline void qualities_head_tail(const DevModel &m, const QualState &Q, const ReadOut &ro, const ns_key &key, uint32_t a,
                                           uint32_t head, uint32_t tail, uint32_t lane) {
    const uint32_t total = head + tail;
    for (uint32_t j = lane; 16u * j < total; j += 64) {
        const uint32_t cs[4] = {0x08080808u * (uint32_t)NS_Q_HT, 0x08080808u * (uint32_t)NS_Q_HT, 0x08080808u * (uint32_t)NS_Q_HT, 0x08080808u * (uint32_t)NS_Q_HT};
        uint64_t qlo, qhi;
        quals16(Q, m, key, ST_HTQ, 0, a, j, cs, qlo, qhi);
        const uint32_t m0 = 16u * j, m1 = min(m0 + 16u, total);
        if (m0 < head) store_qual_chunk(ro, m0, min(m1, head) - m0, qlo, qhi);
        if (m1 > head) {
            const uint32_t lo = max(m0, head);
            shift_down_bytes(qlo, qhi, lo - m0);
            store_qual_chunk(ro, ro.seq_len - tail + (lo - head), m1 - lo, qlo, qhi);
        }
    }
}
