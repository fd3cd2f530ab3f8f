// chain_host.hip — TEST infrastructure: the thread-per-read event chains of the engine (nanosim_amd/csrc/ns_chain.h: chain_error_list,
// chain_error_list_g, chain_unaligned_error_list) and the table packing of ns_load_model (ns_pack.h) compiled for the HOST, so that `-m "not gpu"` tests can
// hold the device source against the oracle event by event (tests/test_chain_host.py).  Built by the test with
//   hipcc --cuda-host-only -x hip -O2 -std=c++17 -ffp-contract=off -DNS_HOST_TEST -shared -fPIC
// (host pass only: no device code, no HIP call).  Nothing of the product links or loads this file.
#include <new>
#include "../nanosim_amd/csrc/ns_materialise.h"   // (wave_incl_scan: the cooperative chain of ns_chain.h, same include order as the engine)
#include "../nanosim_amd/csrc/ns_chain.h"
#include "../nanosim_amd/csrc/ns_pack.h"

struct ChainHost {
    ChainTab ct;
    std::vector<uint64_t> blob;
    bool whole;
};

extern "C" {

void *chost_pack(const ns_model_tables *t) {
    ChainHost *h = new (std::nothrow) ChainHost;
    if (!h) return nullptr;
    ns_pack_chain_tables(t, t->mm_seg_off[t->mm_nbins], h->ct, h->blob, h->whole);
    return h;
}
void chost_free(void *p) { delete static_cast<ChainHost *>(p); }
int chost_whole(const void *p) { return static_cast<const ChainHost *>(p)->whole ? 1 : 0; }
uint32_t chost_lds_words(const void *p) { return static_cast<const ChainHost *>(p)->ct.n_words_lds; }
uint32_t chost_tail_bits(const void *p) { return static_cast<const ChainHost *>(p)->ct.tail_bits; }

// variant 0: chain_error_list   — the integer image k_chain<LDS> walks (T = the blob's first n_words_lds words)
//         1: chain_error_list_g — the fp64 tables (models whose value edges are not whole numbers, tables too large for LDS)
//         2: chain_unaligned_error_list on the LDS image, 3: on the whole blob (k_chain<false, .>)
//         4: chain_error_list with the whole blob as its image (an integer image that does not fit LDS)
// T and TG: the LDS image is the first n_words_lds words of the blob — handing the chain a COPY of just those words as T checks that it
// never reads a table of the LDS part behind them.
// staged != 0: events go through the four-slot staging column (EvSink32::stg) as in k_chain<LDS> for single-piece reads; cap must then be
// a multiple of four and ev 32-byte aligned.  Returns 0, or -1 for an unknown variant.
int chost_error_list(const void *p, int variant, int staged, int32_t m_ref, uint64_t seed, uint64_t read, uint32_t seg, uint32_t attempt,
                     ns_event *ev, uint32_t cap, int32_t *l_new, int32_t *middle_ref, uint32_t *n_ev, int32_t *shift, int *overflow, int *range) {
    const ChainHost *h = static_cast<const ChainHost *>(p);
    const Tabs TG{h->blob.data()};
    std::vector<uint64_t> lds(h->blob.begin(), h->blob.begin() + h->ct.n_words_lds);      // (+ nothing: a read behind it is out of bounds)
    const Tabs T{lds.data()};
    const ns_key key{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)read, (uint32_t)(read >> 32)};
    uint2 stage[4 * NS_CHAIN_BLOCK];
    EvSink32 s;
    s.ev = ev; s.cap = cap; s.n = 0; s.shift = 0; s.last_ins_len = 0; s.overflow = false; s.range = false;
    s.stg = staged ? stage : nullptr; s.stride = NS_CHAIN_BLOCK;
    EList32 e;
    switch (variant) {
    case 0: e = chain_error_list(T, TG, h->ct, m_ref, key, seg, attempt, s); break;
    case 1: e = chain_error_list_g(TG, h->ct, m_ref, key, seg, attempt, s); break;
    case 2: e = chain_unaligned_error_list(T, h->ct, m_ref, key, seg, attempt, s); break;
    case 3: e = chain_unaligned_error_list(TG, h->ct, m_ref, key, seg, attempt, s); break;
    case 4: e = chain_error_list(TG, TG, h->ct, m_ref, key, seg, attempt, s); break;     // the integer image read from global memory (k_chain<false, false>)
    default: return -1;
    }
    ev_flush_tail(s);
    *l_new = e.l_new; *middle_ref = e.middle_ref; *n_ev = s.n; *shift = s.shift; *overflow = s.overflow; *range = s.range;
    return 0;
}

}  // extern "C"
