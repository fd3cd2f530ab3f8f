// ns_hp.h — -k/--KmerBias: the homopolymer filter of mutate_read (S:1920-1947) and mutate_homo (S:618-705).
//
// Lengths change AFTER the mutated segment exists, so the mode is count-then-write:
//   k_hp_filter   thread/read  drop the events that overlap a homopolymer of the un-mutated segment, re-pack the rest
//   k_materialise              (unchanged kernel) writes the pre-homopolymer read, forward strand, into a scratch buffer
//   k_hp_count    thread/read  scan the scratch segments for runs >= k, draw the new run lengths -> final lengths,
//                              final length check (S:1429); a failing read bumps its attempt state and the batch is re-run
//   k_hp_write    thread/read  scratch -> final record with the runs re-sampled, mismatches, qualities, revcomp
// First implementation: one thread per read for the three hp kernels (sequential like the reference); the main
// path keeps its wave-per-read kernels.
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

// get_nd_par (src/model_homopolymer_lengths.py:246-260)
__device__ __forceinline__ void hp_nd_par(const DevModel &m, uint32_t base, uint32_t len, double &mu, double &sigma) {
    const ns_hp_class &h = m.hp[(base == 'A' || base == 'T') ? 0 : 1];
    const double x = (double)len;
    double y = h.konst + h.alpha1 * x;
    for (uint32_t j = 0; j < h.n_breaks; ++j) {
        const double d = x - h.breakpoint[j];
        y += h.beta[j] * (d > 0 ? d : 0.0);
    }
    mu = y; sigma = h.intercept + h.slope * x;
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
// one base of a re-sampled run (S:671-682)
__device__ __forceinline__ uint8_t hp_base(const DevModel &m, uint32_t base, const ns_key &key, uint32_t sid, uint32_t a,
                                           uint32_t idx, uint32_t sub, bool &is_mis) {
    u32x4 w = ns_draw(key, ST_HPMIS, sid, a, idx, sub);
    const double p = u32_to_p(w.x);
    if (!(0 < p && p <= m.hp_mis_rate)) { is_mis = false; return (uint8_t)base; }
    const uint32_t j = (uint32_t)(((uint64_t)w.y * 3u) >> 32);
    const int rc = base_rank(base);
    is_mis = true;
    return bases_atcg(j + ((int)j >= rc ? 1u : 0u));
}
