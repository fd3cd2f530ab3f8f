// Test shim (CPU tests only): the engine's one-pass walk over cs strings (nanosim_amd/csrc/ns_cs_hist.h — the code k_cs_hist runs per
// thread) compiled for the HOST with a plain accumulator, so that the walk itself is checked against the oracle's two-list restatement
// and the reference's files without a GPU.  Built by tests/test_characterize.py with g++ into tests/_tmp/.
#include <stdint.h>
#include <string.h>
#include "../nanosim_amd/csrc/ns_cs_hist.h"

struct HostAcc {
    uint64_t *dic, *mat, *errs, *firsts, *misc; uint32_t cap2;
    void d1(uint32_t which, uint32_t v) { if (v <= NS_CS_DICT_MAX) dic[which * 1001u + v] += 1; }
    void m2(uint32_t p, uint32_t s) {
        const uint32_t mx = p > s ? p : s;
        if (mx > misc[0]) misc[0] = mx;
        if (mat && mx < cap2) mat[(uint64_t)p * cap2 + s] += 1; else misc[1] += 1;
    }
    void err(uint32_t i) { errs[i] += 1; }
    void first(uint32_t i) { firsts[i] += 1; }
    void skip() { misc[2] += 1; }
};

extern "C" int csh_host_count(const uint8_t *cs, const uint64_t *off, uint32_t n_aln, uint32_t cap2, uint64_t *dic, uint64_t *m2,
                              uint64_t *err, uint64_t *first, uint64_t *misc) {
    // per alignment exactly what k_cs_hist does per thread (independent alignments: the carry is looked up, not threaded through)
    for (uint32_t a = 0; a < n_aln; ++a) {
        const uint8_t *s = cs + off[a];
        const uint64_t n = off[a + 1] - off[a];
        uint32_t pm = 0;
        { CsCursor c; cs_cursor_init(c); int t; uint32_t l; if (cs_next_op(s, n, c, t, l) && t != CS_MATCH) pm = cs_carry_in(cs, off, a); }
        HostAcc acc{dic, m2, err, first, misc, cap2};
        cs_hist_alignment(s, n, pm, (bool *)0, acc);
    }
    return 0;
}

// the MAF branch (maf_hist_alignment): the two lines of every alignment at the same offsets
extern "C" int maf_host_count(const uint8_t *ref, const uint8_t *qry, const uint64_t *off, uint32_t n_aln, uint32_t cap2, uint64_t *dic, uint64_t *m2,
                              uint64_t *err, uint64_t *first, uint64_t *misc) {
    for (uint32_t a = 0; a < n_aln; ++a) {
        HostAcc acc{dic, m2, err, first, misc, cap2};
        maf_hist_alignment(ref + off[a], qry + off[a], off[a + 1] - off[a], acc);
    }
    return 0;
}
