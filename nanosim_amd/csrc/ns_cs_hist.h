// ns_cs_hist.h — training-side histogramming (second half): what src/besthit_to_histogram.py:hist() (B:148-486)
// counts from the cs strings of the primary alignments — match / mismatch / insertion / deletion run lengths, the first match of every
// alignment, the (previous match, next match) matrix behind _match_markov_model, and the error transitions behind _error_markov_model —
// as ONE streaming pass per alignment.  B: = src/besthit_to_histogram.py of bcgsc/NanoSim v3.2.2.
//
// The reference works in two steps: parse_cs (B:41-69) turns the string into two lists (one entry per op, a run of `*xy` items folded
// into one "mis" op that carries its count), then hist() walks the list with a little state (B:316-365): `flag` (no error seen yet in
// this alignment), prev_error, prev_match, and two looks sideways — the op BEFORE an error (Python's list[i - 1]: for the first op
// that is the LAST op of the alignment) and whether a match is the last op.  Both looks are local, so the walk streams: the items are
// tokenised on the fly (the regex of B:46), folded, and an op is accounted for when the op behind it is known.
// The code below compiles for the device (k_cs_hist) and, unchanged, for the host (tests/cs_hist_host.cpp: the CPU tests run the very
// same walk against the oracle's two-list restatement and the reference's files).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define NS_CSH __host__ __device__ __forceinline__
#else
#define NS_CSH static inline
#endif

enum { CS_MATCH = 0, CS_MIS = 1, CS_INS = 2, CS_DEL = 3, CS_SKIP = 4 };         // conv_op_to_word (B:135-145)
enum { CSH_MATCH = 0, CSH_FIRST = 1, CSH_MIS = 2, CSH_INS = 3, CSH_DEL = 4 };    // the five 1-D histograms (add_dict, B:14-22)
#define NS_CS_DICT_MAX 1000u      // add_dict ignores values above it (B:15-16)

// Where the bytes of a string come from: `const uint8_t *` (host, MAF branch) or CsBytes (k_cs_hist) — the walks only index it.
// CsBytes keeps the aligned 8 bytes around the last position in a register pair: a thread walks its own string, so a byte load of a
// wavefront touches 64 cache lines, and that, not the arithmetic, is what the kernel waits for (one load per 8 bytes instead of one per
// byte; the strings are allocated with 16 bytes to spare, so the last window may reach beyond a string's end).
#if defined(__HIPCC__)
struct CsBytes {
    const uint8_t *p;
    uint64_t w; uintptr_t at;
    __device__ __forceinline__ explicit CsBytes(const uint8_t *q) : p(q), w(0), at(~(uintptr_t)0) {}
    __device__ __forceinline__ uint8_t operator[](uint64_t i) {
        const uintptr_t a = (uintptr_t)p + i, al = a & ~(uintptr_t)7;
        if (al != at) { w = *reinterpret_cast<const uint64_t *>(al); at = al; }
        return (uint8_t)(w >> (8u * (uint32_t)(a & 7u)));
    }
};
#endif

// the next item of the cs string at or after i — re.findall('(:[0-9]+|\*[a-z][a-z]|[=\+\-][A-Za-z]+)') of B:46: what matches nothing
// is skipped one character at a time
template <class S>
NS_CSH bool cs_next_item(S &s, uint64_t n, uint64_t &i, int &type, uint32_t &len) {
    while (i < n) {
        const uint8_t c = s[i];
        if (c == ':') {
            uint64_t j = i + 1, v = 0;
            while (j < n && s[j] >= '0' && s[j] <= '9') { v = v * 10 + (uint64_t)(s[j] - '0'); if (v > 0xffffffffull) v = 0xffffffffull; ++j; }
            if (j > i + 1) { type = CS_MATCH; len = (uint32_t)v; i = j; return true; }
        } else if (c == '*') {
            if (i + 2 < n && s[i + 1] >= 'a' && s[i + 1] <= 'z' && s[i + 2] >= 'a' && s[i + 2] <= 'z') { type = CS_MIS; len = 1; i += 3; return true; }
        } else if (c == '+' || c == '-' || c == '=') {
            uint64_t j = i + 1;
            while (j < n && ((s[j] >= 'a' && s[j] <= 'z') || (s[j] >= 'A' && s[j] <= 'Z'))) ++j;
            if (j > i + 1) { type = c == '+' ? CS_INS : c == '-' ? CS_DEL : CS_SKIP; len = (uint32_t)(j - i - 1); i = j; return true; }
        }
        ++i;
    }
    return false;
}

// the next op of the FOLDED list (list_op / list_hist of parse_cs): a run of mismatch items is one op whose length is their number
struct CsCursor { uint64_t i; bool have; int type; uint32_t len; };
NS_CSH void cs_cursor_init(CsCursor &c) { c.i = 0; c.have = false; c.type = CS_SKIP; c.len = 0; }
template <class S>
NS_CSH bool cs_next_op(S &s, uint64_t n, CsCursor &c, int &type, uint32_t &len) {
    int t; uint32_t l;
    if (!c.have) { if (!cs_next_item(s, n, c.i, t, l)) return false; c.have = true; c.type = t; c.len = l; }
    type = c.type; len = c.len;
    c.have = false;
    if (type == CS_MIS) {                                       // fold the mismatch items that follow (B:49-52, 64-65)
        while (cs_next_item(s, n, c.i, t, l)) {
            if (t != CS_MIS) { c.have = true; c.type = t; c.len = l; break; }
            ++len;
        }
    }
    return true;
}

// the walk of hist() over one alignment (B:328-365).  prev_match: in = the value the previous alignments left (the reference never
// resets it), out = what this one leaves; *assigned: the alignment assigned it.  n_skip counts `=` items (long-form cs): the reference's
// two lists fall out of step on them, the caller refuses such input.
template <class Acc, class S>
NS_CSH void cs_hist_alignment(S &s, uint64_t n, uint32_t &prev_match, bool *assigned, Acc &acc) {
    CsCursor c; cs_cursor_init(c);
    bool flag = true;
    int prev_error = CS_MIS;                                    // (only read after an error of this alignment has set it)
    int t; uint32_t l;
    bool more = cs_next_op(s, n, c, t, l);
    // the type of the last op: what list_op_unique[i - 1] is for i = 0 (B:332) — looked at only when the FIRST op is an error, so only
    // then is the string walked for it (until round 6 every alignment was walked twice)
    int prev_type = CS_SKIP;
    if (more && t != CS_MATCH && t != CS_SKIP) { CsCursor c2; cs_cursor_init(c2); int t2; uint32_t l2; while (cs_next_op(s, n, c2, t2, l2)) prev_type = t2; }
    while (more) {
        int tn = CS_SKIP; uint32_t ln = 0;
        const bool has_next = cs_next_op(s, n, c, tn, ln);
        if (t == CS_SKIP) acc.skip();
        else if (t != CS_MATCH) {                               // B:331-352
            const bool zero = prev_type != CS_MATCH;            // exact_prev_op != "match": prev_error += "0"
            if (flag) { flag = false; acc.first((uint32_t)(t - CS_MIS)); }
            else acc.err((uint32_t)((prev_error - CS_MIS) + (zero ? 3 : 0)) * 3u + (uint32_t)(t - CS_MIS));
            prev_error = t;
            if (t == CS_MIS) {
                acc.d1(CSH_MIS, l);
                if (zero) { acc.d1(CSH_MATCH, 0); acc.m2(prev_match, 0); prev_match = 0; if (assigned) *assigned = true; }
            } else if (t == CS_DEL) acc.d1(CSH_DEL, l);
            else acc.d1(CSH_INS, l);
        } else {                                                // B:353-364
            if (flag) { acc.d1(CSH_FIRST, l); prev_match = l; if (assigned) *assigned = true; }
            else if (!has_next) acc.m2(prev_match, l);
            else { acc.d1(CSH_MATCH, l); acc.m2(prev_match, l); prev_match = l; if (assigned) *assigned = true; }
        }
        if (t != CS_SKIP) prev_type = t;                        // (a `skip` op is not looked at by conv_op_to_word's callers: B:330-332)
        else prev_type = CS_SKIP;
        t = tn; l = ln; more = has_next;
    }
}

// an accumulator that counts nothing: what an alignment leaves in prev_match
struct CsAccNull {
    NS_CSH void d1(uint32_t, uint32_t) {}
    NS_CSH void m2(uint32_t, uint32_t) {}
    NS_CSH void err(uint32_t) {}
    NS_CSH void first(uint32_t) {}
    NS_CSH void skip() {}
};
// prev_match in front of alignment a: the last value an earlier alignment assigned (0 in front of all of them: the reference would
// stop with an UnboundLocalError there)
NS_CSH uint32_t cs_carry_in(const uint8_t *cs, const uint64_t *off, uint64_t a) {
    while (a > 0) {
        --a;
        uint32_t pm = 0; bool assigned = false; CsAccNull z;
#if defined(__HIP_DEVICE_COMPILE__)
        CsBytes s(cs + off[a]);
#else
        const uint8_t *s = cs + off[a];
#endif
        cs_hist_alignment(s, off[a + 1] - off[a], pm, &assigned, z);
        if (assigned) return pm;
    }
    return 0;
}

// ---- the MAF branch of hist() (B:188-315): the two aligned lines of an alignment, column by column ------------------------------------
// Same counters, another tokenizer: a column is a match (equal letters after upper()), an inserted base (reference '-'), a deleted base
// (query '-') or a mismatch.  The reference keeps FOUR pending run lengths and flushes at most one of them per column through its
// `elif` chains — so a deletion directly in front of an insertion leaves both pending, and the one that is tested later waits until the
// earlier one has gone; prev_match and prev_error start afresh in every alignment; runs still pending behind the last column are
// dropped, and a match that ends the alignment goes to match_list only (B:233-234).  All of that is the walk below.
struct MafRuns { uint32_t match, mis, ins, del; };
template <class Acc>
NS_CSH void maf_hist_alignment(const uint8_t *ref, const uint8_t *qry, uint64_t n, Acc &acc) {
    MafRuns run{0u, 0u, 0u, 0u};
    uint32_t prev_match = 0, state = 0;           // state: the row of error_list the next error is counted in (mis ins del mis0 ins0 del0)
    bool first = true;
    auto upper = [](uint8_t c) -> uint8_t { return (c >= 'a' && c <= 'z') ? (uint8_t)(c - 32) : c; };
    // an error run of kind t (0 mis, 1 ins, 2 del) is over; zero: no match between it and the column that ends it
    auto close_error = [&](uint32_t t, uint32_t len, bool zero) {
        acc.d1(CSH_MIS + t, len);
        if (zero) { acc.d1(CSH_MATCH, 0); acc.m2(prev_match, 0); prev_match = 0; }
        if (first) { first = false; acc.first(t); } else acc.err(state * 3u + t);
        state = t + (zero ? 3u : 0u);
    };
    auto close_match = [&]() {
        if (first) acc.d1(CSH_FIRST, run.match);
        else { acc.d1(CSH_MATCH, run.match); acc.m2(prev_match, run.match); }
        prev_match = run.match; run.match = 0;
    };
    for (uint64_t i = 0; i < n; ++i) {
        const uint8_t r = upper(ref[i]), q = upper(qry[i]);
        if (r == q) {                                                   // B:204-234
            if (run.mis) { close_error(0, run.mis, false); run.mis = 0; }
            else if (run.ins) { close_error(1, run.ins, false); run.ins = 0; }
            else if (run.del) { close_error(2, run.del, false); run.del = 0; }
            ++run.match;
            if (i + 1 == n) acc.m2(prev_match, run.match);
        } else if (r == '-' || q == '-') {                              // B:235-280: an indel column ends a match or a mismatch run
            if (run.match) close_match();
            else if (run.mis) { close_error(0, run.mis, true); run.mis = 0; }
            if (r == '-') ++run.ins; else ++run.del;
        } else {                                                        // B:281-315: a mismatch column ends a match or an indel run
            if (run.match) close_match();
            else if (run.ins) { close_error(1, run.ins, true); run.ins = 0; }
            else if (run.del) { close_error(2, run.del, true); run.del = 0; }
            ++run.mis;
        }
    }
}
