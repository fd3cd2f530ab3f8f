// errlog_host.hip — TEST infrastructure: the error-profile row writer of -DNS_ERRLOG_V3 (nanosim_amd/csrc/ns_errlog.h: errlog_tail_v3, the
// packed 8-byte stores) compiled for the HOST and run in the order k_errlog runs it — per block of 64 events the row tails of all lanes,
// then the read names over what the tails spilled into them, then the block leaves — so that `-m "not gpu"` tests can compare the bytes
// with the oracle's error profile (tests/test_errlog_host.py).  Built with
//   hipcc --cuda-host-only -x hip -O2 -std=c++17 -ffp-contract=off -DNS_HOST_TEST -shared -fPIC
// Nothing of the product links or loads this file.
#include <string.h>
#include <vector>
#include "../nanosim_amd/csrc/ns_errlog.h"

extern "C" {

// returns the bytes written to `out`, -1: out too small, -2: a block that k_errlog would not stage (row block > 8 192 bytes, name < 8 or
// > 256 bytes: those take the round-4 path)
int64_t elhost_errlog(const ns_read *reads, uint64_t n_reads, const ns_piece *pieces, const ns_event *events, const uint8_t *records,
                      const uint32_t *name_len, const uint8_t *bases_raw, uint64_t n_bases, const uint64_t *chrom_off, uint32_t nchrom,
                      uint64_t seed, uint64_t first_read, uint8_t *out, uint64_t cap) {
    std::vector<uint8_t> padded(n_bases + 2 * 64, (uint8_t)'N');
    for (uint64_t i = 0; i < n_bases; ++i) padded[64 + i] = normalise_base(bases_raw[i]);      // the engine's copy: normalised, padded (NS_REF_PAD)
    DevRef ref; memset(&ref, 0, sizeof ref);
    ref.bases = padded.data() + 64; ref.chrom_off = chrom_off; ref.nchrom = nchrom;
    std::vector<uint8_t> lds(16 + 8192 + 16);
    uint8_t *const buf = lds.data() + 16;
    uint64_t base = 0;
    for (uint64_t r = 0; r < n_reads; ++r) {
        const ns_read rd = reads[r];
        if (rd.flags) continue;
        const uint64_t g = first_read + r;
        const ns_key key{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)g, (uint32_t)(g >> 32)};
        const uint32_t a = rd.attempts, nl = name_len[r];
        const uint8_t *name = records + rd.rec_off + 1;
        if (nl < 8 || nl > 256) return -2;
        for (uint32_t pi = 0; pi < rd.n_pieces; pi += 2) {
            const ns_piece p = pieces[rd.piece_off + pi];
            const PieceCtx pc = load_piece(events, ref, p, pi);
            for (uint32_t j0 = 0; j0 < p.n_ev; j0 += 64) {
                uint32_t off[65]; off[0] = 0;
                ns_event ev[64];
                uint32_t n_act = 0;
                for (uint32_t lane = 0; lane < 64; ++lane) {
                    const uint32_t k = j0 + lane;
                    uint32_t row = 0;
                    if (k < p.n_ev) {
                        ev[lane] = pc.ev[p.n_ev - 1 - k];                       // rows are written from the LAST event to the first
                        const uint32_t len = ns_ev_len(ev[lane].info);
                        row = nl + dec_digits(ev[lane].pos) + dec_digits(len) + 2u * len + 9u;
                        n_act = lane + 1;
                    }
                    off[lane + 1] = off[lane] + row;
                }
                const uint32_t total = off[64];
                if (total > 8192u) return -2;
                memset(lds.data(), 0xee, lds.size());
                for (uint32_t lane = 0; lane < n_act; ++lane) {              // 1. the tails of all rows
                    const uint32_t k = j0 + lane;
                    const uint32_t tl = errlog_tail_v3(buf + off[lane] + nl, ev[lane], p.n_ev - 1 - k, pc, ref, key, a);
                    if (tl + nl != off[lane + 1] - off[lane]) return -3;
                }
                for (uint32_t lane = 0; lane < n_act; ++lane) {              // 2. the names, over what the tails spilled into them
                    uint8_t *q = buf + off[lane];
                    for (uint32_t i = 0; i + 8 <= nl; i += 8) memcpy(q + i, name + i, 8);
                    if (nl & 7u) memcpy(q + nl - 8, name + nl - 8, 8);
                }
                if (base + total > cap) return -1;
                memcpy(out + base, buf, total);                              // 3. the block leaves
                base += total;
            }
        }
    }
    return (int64_t)base;
}

// the row tail of ONE insertion of `len` letters at reference position `pos` (no reference base is read): for the number formats
int32_t elhost_ins_tail(uint32_t pos, uint32_t len, uint8_t *out /* >= 2 len + 40 bytes */) {
    ns_event e; e.pos = pos; e.info = ns_ev_pack(len, NS_INS, 0);
    PieceCtx pc; memset(&pc, 0, sizeof pc);
    pc.chrom_len = ~0ull;
    DevRef ref; memset(&ref, 0, sizeof ref);
    const ns_key key{1u, 2u, 3u, 0u};
    return (int32_t)errlog_tail_v3(out, e, 0u, pc, ref, key, 0u);
}

}  // extern "C"
