// ns_errlog.h — one row of the error profile (out_error.write, S:2006-2008):
//   <read name> \t <position> \t mis|ins|del \t <length> \t <reference bases or -...> \t <new bases or -...> \n
// written with 8-byte stores (-DNS_ERRLOG_V3; prepared in round 4, held against the oracle on the CPU by tests/test_errlog_host.py, not yet
// timed).  k_errlog assembles the rows of 64 events side by side in an LDS block; its round-4 form writes every byte of a row's tail on
// its own (put_dec, the letters: one ds_write_b8 and its address arithmetic per byte, ≈1 500 instructions per 64 rows).  Here a field is
// packed in a 64-bit register and leaves as ONE unaligned 8-byte store of which the first n bytes count: the bytes behind them are
// overwritten by the next field's store.  The last store of a row may run up to 7 bytes into the NEXT row of the block — into its read
// name, which k_errlog therefore writes AFTER the tails (behind a wavefront barrier; names are at least 8 bytes, else the old path).
#pragma once
#include "ns_materialise.h"

NS_DEV uint8_t *el_emit(uint8_t *w, uint64_t word, uint32_t n) { __builtin_memcpy(w, &word, 8); return w + n; }
// the n = dec_digits(v) digits of v < 10^8 as bytes, the most significant in byte 0
NS_DEV uint64_t el_dec8(uint32_t v, uint32_t n) {
    uint64_t acc = 0;
    for (uint32_t i = 0; i < n; ++i) { const uint32_t q = v / 10u; acc = (acc << 8) | (uint64_t)('0' + (v - 10u * q)); v = q; }
    return acc;
}
NS_DEV uint64_t el_low_bytes(uint64_t w, uint32_t n) { return n >= 8u ? w : w & ((1ull << (8u * n)) - 1ull); }

// the row of event e (index j of its piece) behind the read name, at q; returns its length = dec_digits(pos) + dec_digits(len) + 2 len + 9
NS_DEV uint32_t errlog_tail_v3(uint8_t *q, const ns_event &e, uint32_t j, const PieceCtx &pc, const DevRef &ref, const ns_key &key, uint32_t a) {
    uint8_t *w = q;
    const uint32_t len = ns_ev_len(e.info), ty = ns_ev_type(e.info), pos = e.pos;
    if (pos >= 100000000u) {                                                               // "\t<position>"
        const uint32_t hi = pos / 100000000u, lo = pos - hi * 100000000u, nh = hi < 10u ? 1u : 2u;
        w = el_emit(w, (uint64_t)'\t' | el_dec8(hi, nh) << 8, 1u + nh);
        w = el_emit(w, el_dec8(lo, 8u), 8u);
    } else {
        const uint32_t n1 = dec_digits(pos);
        const uint64_t d = el_dec8(pos, n1);
        if (n1 <= 7u) w = el_emit(w, (uint64_t)'\t' | d << 8, 1u + n1);
        else { w = el_emit(w, (uint64_t)'\t', 1u); w = el_emit(w, d, 8u); }
    }
    const uint64_t tn = ty == NS_MIS ? 0x0973696d09ull : ty == NS_INS ? 0x09736e6909ull : 0x096c656409ull;   // "\tmis\t" "\tins\t" "\tdel\t"
    const uint32_t n2 = dec_digits(len);
    const uint64_t d2 = el_dec8(len, n2);
    if (n2 <= 2u) w = el_emit(w, tn | d2 << 40 | (uint64_t)'\t' << (40u + 8u * n2), 6u + n2);          // "\t<type>\t<length>\t"
    else { w = el_emit(w, tn, 5u); w = el_emit(w, d2 | (uint64_t)'\t' << (8u * n2), n2 + 1u); }
    uint8_t *w2 = w + len + 1u;
    // the reference bases under a substitution / deletion of <= 8 bases: ONE 8-byte load (across the origin, or longer: byte loads)
    uint64_t ref8 = 0;
    const bool ref_fast = ty != NS_INS && len <= 8u && pc.pos + pos + 8ull <= pc.chrom_len;
    if (ref_fast) __builtin_memcpy(&ref8, ref.bases + pc.chrom_base + pc.pos + pos, 8);
    auto cur_at = [&](uint32_t i) -> uint32_t {
        const uint32_t x = pos + i;
        return resolve_base(ref_fast ? (uint32_t)(ref8 >> (8u * i)) & 0xffu : (uint32_t)ref_base_at(ref, pc, x), key, pc.sid, a, x);
    };
    // column 5: the bases of the reference ('-' under an insertion), then the tab.  ALL its stores come before those of column 6: the
    // last one runs over the start of column 6
    for (uint32_t c = 0; c < len; c += 8u) {
        const uint32_t n = len - c < 8u ? len - c : 8u;
        uint64_t word = 0x2d2d2d2d2d2d2d2dull;
        if (ty != NS_INS) { word = 0; for (uint32_t i = 0; i < n; ++i) word |= (uint64_t)cur_at(c + i) << (8u * i); }
        if (n < 8u) w = el_emit(w, el_low_bytes(word, n) | (uint64_t)'\t' << (8u * n), n + 1u);
        else w = el_emit(w, word, 8u);
    }
    if (!(len & 7u)) w = el_emit(w, (uint64_t)'\t', 1u);
    // column 6: the new bases ('-' under a deletion), then the end of the line.  The letters of an event come from one word per 16:
    // 2-bit fields (insertion), successive base-3 digits (substitution) — payload_word
    uint32_t frac = 0;
    for (uint32_t c = 0; c < len; c += 8u) {
        const uint32_t n = len - c < 8u ? len - c : 8u;
        uint64_t word = 0x2d2d2d2d2d2d2d2dull;
        if (ty != NS_DEL) {
            word = 0;
            for (uint32_t i = 0; i < n; ++i) {
                const uint32_t k = c + i;
                if (!(k & 15u)) frac = payload_word(key, pc.sid, a, j, k >> 4);
                const uint32_t b = ty == NS_INS ? (uint32_t)bases_atcg((frac >> (2u * (k & 15u))) & 3u) : (uint32_t)mis_from_digit(cur_at(k), next_digit3(frac));
                word |= (uint64_t)b << (8u * i);
            }
        }
        if (n < 8u) w2 = el_emit(w2, el_low_bytes(word, n) | (uint64_t)'\n' << (8u * n), n + 1u);
        else w2 = el_emit(w2, word, 8u);
    }
    if (!(len & 7u)) w2 = el_emit(w2, (uint64_t)'\n', 1u);
    return (uint32_t)(w2 - q);
}
