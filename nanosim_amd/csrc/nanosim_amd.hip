// nanosim_amd.hip — gfx950 kernels + C-ABI host layer of the read-generation engine.
//
// Pipeline of one ns_generate() call (one worker call of the reference, S:1266-1454 / S:1482-1549):
//   k_nseg, k_lengths   thread/read   segment count (S:1276-1299), lengths of the attempt, strand, head / tail, event capacity
//   scans, k_order_*    rocPRIM / own piece / event offsets; visiting order by descending planned work (1 024 bins; reads of several pieces first)
//   k_chain             thread/read   error_list / unaligned_error_list, acceptance, positions   (S:1283-1402, 1833-1916, 1784-1830,
//                       (+ wave/read for the longest reads, the unaligned reads and the gaps)      1694-1781); one pass per attempt
//                       (+ thread/PIECE for the reads of several pieces of a chimeric batch: piece modes 1 + 2, GenArgs.piece_mode)
//   k_stats_fold        1 wave        the chain kernels' counters: 64 copies -> one (one set of counters serialises their atomics)
//   k_ir_splice         wave/read     transcriptome: retained introns spliced into the read's slot of an arena (S:1156-1192)
//   -k: k_hp_filter_w, k_materialise<., MAT_HP_SCRATCH>, k_hp_scan, k_hp_drain, k_hp_finalize  (ns_hp.h; S:1920-1947, 618-705)
//   scans               rocPRIM       record / error-profile offsets
//   k_names             thread/read   ">name\n", "+\n" framing                              (S:1390-1402, 1437-1443)
//   k_materialise       wave/read     case_convert + mutate_read + head/tail + revcomp (+ qualities)   (S:743-755, 1919-2015, 1421-1435)
//   k_materialise_dense wave/segment  the same for unaligned reads and gaps (0.55 events per base)
//   k_errlog            wave/read     _aligned_error_profile rows                           (S:2006-2008)
// Metagenome worker calls run k_lengths / k_chain per PASS of the reference's while loop (S:844-1040) with k_meta_* around them; the lists of a
// pass are launched before the host has walked the species quotas (assign_species), k_meta_tail does positions + acceptance afterwards.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "ns_device.h"
#include "ns_materialise.h"
#include "ns_chain.h"
#include "ns_hp.h"
#include "ns_ir.h"
#include "ns_io.h"
#include "ns_pack.h"
#include "ns_cs_hist.h"

// Reads per workgroup of the wave-per-read kernels.  One: read lengths vary by an order of magnitude inside a batch, and a wavefront
// that is done cannot leave before the longest read of its workgroup is.
#ifndef NS_WPB
#define NS_WPB 1
#endif

// ---------------------------------------------------------------------------------------------------------
// kernel arguments
// ---------------------------------------------------------------------------------------------------------
struct GenArgs {
    uint32_t ev_stage;               // k_chain<LDS>: byte offset of the event staging area behind the tables in dynamic LDS (0: none)
    uint32_t hole_at, hole_len;      // k_chain over a list with a hole: entries [hole_at, hole_at + hole_len) belong to another launch (the wave-per-read list of a
                                     // chimeric batch: the longest single-segment reads, which sit behind the reads of several pieces in the visiting order)
    uint32_t piece_mode;             // k_chain, thread per read, pass 0 of a chimeric batch: 1 = a thread per PIECE of the reads of several pieces (slot s of
                                     // NS_PIECE_SLOTS takes piece s, the last slot all pieces from there on): the error list only, left in the piece record;
                                     // 2 = the reads themselves, their lists done (acceptance, positions, name); 0 = a read's lists and the rest in one thread
    const uint32_t *prio_thr;        // k_chain, thread per read, over a batch with reads of several pieces: [1..3] = the planned work from which a wavefront takes
                                     // issue priority 3, 2, 1 (k_order_scan; nullptr: by position in the list, which is then one descending order)
    uint32_t defer_tail;             // metagenome pass: k_chain stops in front of the positions (the species are not known yet: the host is still walking the
                                     // quotas of assign_species, S:758-811, while the lists run) and marks the read pending; k_meta_tail goes on from there
    uint32_t coop_k1;                // wave-per-read unaligned chain with one iteration per lane (NS_UCOOP_K=1: the form until round 6)
    uint32_t coop_mix;               // k_chain<false, true>: the launch carries n_words_mix words of dynamic LDS for the front of the blob
    ns_params prm;
    DevModel m;
    DevRef ref;
    DevTrx tx;                  // transcriptome batches (prm.trx)
    uint16_t *polya;            // transcriptome: polyA tail length per read
    DevIr ir;                   // transcriptome with intron retention (prm.model_ir)
    uint64_t *ir_need;          // per read: bytes of its slot in the splice arena (0: not spliced)
    double cap_rate;            // event capacity per aligned reference base
    uint32_t cap_gap_mul;       // event capacity per gap/unaligned reference base
    // per-read planning arrays (n+1)
    uint32_t *n_pieces;         // in: count, after scan: offsets (separate array piece_off)
    uint32_t *piece_off;
    uint64_t *ev_cap;
    // attempts > 0 draw new lengths: every pass gets a fresh region of the event buffer, planned for the reads it visits
    uint64_t *l_cap;            // [list position] capacity of the read in this pass
    const uint64_t *l_off;      // exclusive scan of l_cap (nullptr in pass 0: ev_off[r] is used)
    uint64_t l_base;            // first event slot of this pass's region
    const uint32_t *p_need, *p_off;   // passes > 0 of a chimeric batch (k_replan): pieces a re-planned read needs / exclusive scan
    uint32_t p_base;                  // first piece slot of this pass's region
    uint64_t *ev_off;
    uint64_t *rec_len;
    uint64_t *rec_off;
    uint64_t *err_len;
    uint64_t *err_off;
    uint16_t *name_len;
    uint32_t *sort_key, *sort_idx;   // k_plan: total reference length per read, read index
    const uint32_t *list;            // reads this pass visits (pass 0: sorted by descending length)
    uint32_t list_n, attempt;
    uint32_t list_base;              // list == nullptr: the launch visits reads list_base .. list_base + list_n - 1
    uint32_t *next_list, *next_n;    // reads rejected in this pass
    uint32_t *rstate;                // per read: epoch | consecutive first-check failures << 16
    uint32_t *att_base;              // per read: first attempt number of this run (0 unless the batch is re-run in -k mode)
    uint32_t keep_state;             // k_nseg keeps rstate/att_base (re-run after a failed final length check)
    uint32_t hp;                     // -k active for this batch
    uint32_t dbg;                    // NS_DEBUG_SKIP (profiling only)
    uint32_t errlen_later;           // the error-profile size of a read is computed by k_errlen / k_hp_filter_w, not by k_chain
    uint8_t *scr;                    // -k: the pieces of every read before mutate_homo (forward strand; FASTQ: with their class bits)
    uint64_t *scr_len, *scr_off;     //     bytes per read / exclusive scan
    uint32_t *hp_len;                // -k: final emitted length per piece
    ns_event *hp_ev;                 // -k: the homopolymer edits of every aligned piece as an event list over its scratch bytes
    uint32_t *hp_wd;                 //     (k_hp_drain; format: materialise_piece, MAT_HP_FINAL) and the letter word of every event
    uint32_t *hp_nev;                //     events per piece
    uint32_t hp_shift, hp_pad;       //     capacity of a piece: (scratch bytes >> hp_shift) + hp_pad events (hp_ev_slot)
    uint32_t *hp_pcnt;               //     pieces per read / their exclusive scan: the ordinal of a piece in READ order (piece_off is not:
    const uint32_t *hp_pord;         //     a chimeric read that was planned again lies behind the others)
    // metagenome (one pass = one `while remaining_reads` iteration of S:836-1036)
    uint32_t meta;                   // 0 genome, 1 metagenome
    uint32_t nspecies;
    const uint32_t *species_chrom_off;
    const uint32_t *key_pos;         // per final read: id of its Philox key inside the pass that accepted it (nullptr: the read index)
    uint32_t *key_pos_w;
    uint32_t m_reversed;             // strand of the pass (S:860)
    uint32_t m_passed, m_pieces_passed;   // reads / pieces accepted by earlier passes
    const uint32_t *m_segptr;        // per read position of the pass (+1 sentinel): first entry of its segments in m_len / m_species
    const int32_t *m_len;            // int(round(length)) per assigned segment (S:871)
    const uint16_t *m_species;
    uint64_t ev_base;                // first event slot of this pass in the events buffer
    uint64_t *accept, *accept_scan;  // per pass position: accepted ? 1 | n_pieces << 32 : 0
    ns_read *f_reads; ns_piece *f_pieces; uint16_t *f_name_len; uint64_t *f_rec_len, *f_err_len;   // final (accepted) arrays
    double *draw_x; uint64_t draw_n;
    unsigned long long *species_bases;
    // keys and numbers: read r of a batch draws with the key (seed, key_first + r) and is called name_first + r.  Both are
    // prm.first_read, except in the candidate table of a transcriptome batch (below), whose keys count from the batch's first BLOCK
    uint64_t key_first, name_first;
    // transcriptome, aligned / --perfect batches (S:1080-1104 per block of NS_TRX_BLOCK read indices; k_trx_walk): the candidate table —
    // position i = candidate i % trx_C of block i / trx_C of the batch; 0: no table
    uint32_t trx_C, trx_M;
    const uint32_t *trx_cand;        // [position] pick that became the candidate, 0xffffffff: none
    const uint32_t *trx_pick_e;      // [block * trx_M + pick] transcript (index of the expression view)
    const int32_t *trx_pick_y;       //                        its aligned length under a fresh sample, -1: fails S:1103-1104
    // results
    ns_read *reads;
    ns_piece *pieces;
    ns_event *events;
    uint8_t *records;
    uint8_t *errlog;
    const uint8_t *hp_bm;       // -k: two bits per reference base (k_hp_bitmap), nullptr: none
    uint32_t *cls;              // FASTQ: the class of every base of the aligned pieces, 2 bits each (k_materialise -> k_qualities; cls_word0)
    unsigned long long *stats;  // [0] overflow reads [1] total bases [2] total ref bases [3] events [4] longest accepted read (unaligned batches)
                                // [5] reads that failed the final length check of -k [6] reads queued for the next pass [7] -k event capacity overflow
};

__device__ __forceinline__ ns_key make_key(const GenArgs &A, uint64_t r) {
    uint64_t g = A.key_first + r;
    return ns_key{(uint32_t)A.prm.seed, (uint32_t)(A.prm.seed >> 32), (uint32_t)g, (uint32_t)(g >> 32)};
}
// key of FINAL read r: in metagenome batches the draws of a read are keyed by its position inside the pass that accepted it, in
// transcriptome batches by its candidate (k_trx_commit)
__device__ __forceinline__ ns_key read_key(const GenArgs &A, uint64_t r) { return make_key(A, A.key_pos ? (uint64_t)A.key_pos[r] : r); }
// candidate table of a transcriptome batch: position -> key index (relative to key_first = the first read of the batch's first
// block) and attempt: candidate c of a block draws with the key of read (block start + c mod W), attempt c / W
__device__ __forceinline__ uint64_t trx_key_rel(const GenArgs &A, uint64_t pos) {
    const uint64_t blk = pos / A.trx_C, c = pos % A.trx_C;
    return blk * NS_TRX_BLOCK + c % NS_TRX_BLOCK;
}
__device__ __forceinline__ uint32_t trx_attempt(const GenArgs &A, uint64_t pos) { return (uint32_t)((pos % A.trx_C) / NS_TRX_BLOCK); }

// Segments of a read (S:1276-1277).  The reference draws num_segment once per worker and hands the counts out by POSITION among the reads
// still missing (remaining_segments = num_segment[passed:], S:1447), so a count that no draw of lengths can satisfy is not retried for
// ever.  Here the count belongs to the EPOCH of the read: it is drawn again whenever the read draws new segment lengths.
__device__ __forceinline__ uint32_t read_nseg(const GenArgs &A, const ns_key &key, uint32_t epoch) {
    if (A.prm.kind != NS_KIND_ALIGNED || !A.prm.chimeric) return 1;
    u32x4 w = ns_draw(key, ST_NSEG, 0, epoch, 0, 0);
    uint32_t nseg = (uint32_t)table_value(A.m.nseg_cdf, A.m.nseg_n, u32_to_p(w.x));
    return nseg > NS_MAX_SEG ? NS_MAX_SEG : nseg;
}

// ---------------------------------------------------------------------------------------------------------
// planned work on a scale of 32 steps, four per octave from 2^9 to 2^17 (clamped outside): the visiting order of the reads of several pieces
__device__ __forceinline__ uint32_t ord_coarse(uint64_t w) {
    if (w < 512ull) return 0u;
    if (w >= (1ull << 17)) return 31u;
    const uint32_t v = (uint32_t)w, e = 31u - (uint32_t)__clz((int)v);      // 9 .. 16
    return (e - 9u) * 4u + ((v >> (e - 2u)) & 3u);
}
__device__ __forceinline__ uint32_t ord_coarse_floor(uint32_t b) { return b ? (4u + (b & 3u)) << ((b >> 2) + 7u) : 0u; }   // the smallest work of step b
// k_nseg: pieces per read (S:1276-1279)
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_nseg(GenArgs A) {
    uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > A.prm.n_reads) return;
    if (A.ir_need) A.ir_need[r] = 0;
    if (r == A.prm.n_reads) { A.n_pieces[r] = 0; A.ev_cap[r] = 0; A.rec_len[r] = 0; A.err_len[r] = 0; return; }
    A.n_pieces[r] = 2 * read_nseg(A, make_key(A, r), (A.keep_state && !A.meta) ? A.rstate[r] & 0xffffu : 0u) - 1;
    if (!A.keep_state) { A.rstate[r] = 0; A.att_base[r] = 0; }
}

// passes > 0 of a chimeric batch: a read whose epoch advanced draws its segment count again; if the count changed, the read moves to
// fresh piece slots behind the planned ones (need[tid] = pieces to allocate, 0: it keeps its slots)
__global__ void __launch_bounds__(256) k_replan(GenArgs A, uint32_t *need) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid > A.list_n) return;
    if (tid == A.list_n) { need[tid] = 0; return; }
    const uint64_t r = A.list[tid];
    const uint32_t np = 2 * read_nseg(A, make_key(A, r), A.rstate[r] & 0xffffu) - 1;
    need[tid] = np != A.reads[r].n_pieces ? np : 0u;
}

// ---------------------------------------------------------------------------------------------------------
// k_lengths: everything attempt `a` of a read draws BEFORE its error lists — segment / gap lengths
// (S:1285-1299), head+tail remainder and ratio (S:1471-1474), strand (S:1312).  All the fp64 sampling math
// (KDE, inverse normal, 10^x) lives here; k_chain is integer/table work only.  Pass 0 visits every read and
// also sizes the event buffer; later passes visit only the reads whose previous attempt was rejected.
// ---------------------------------------------------------------------------------------------------------
// SMALL: the retry passes (a handful of reads).  175 VGPRs do not fit between the wavefronts of another call's record kernel (7 x 72 of a
// SIMD's 512): next to it such a launch waited 2 ms for room (round 6, profiles/r06/step_timeline.log).  One wavefront per workgroup within
// 72 registers (the rest spills: irrelevant for a few reads) goes wherever a record wavefront has left.
template <bool SMALL>
__global__ void __launch_bounds__(SMALL ? 64 : 256, SMALL ? 7 : 1) k_lengths(GenArgs A) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= A.list_n) return;
    const uint64_t r = A.list ? A.list[tid] : tid;
    const ns_params &prm = A.prm;
    const int kind = (int)prm.kind;
    const bool meta_al = A.meta && kind != NS_KIND_UNALIGNED;    // r is then the position of the read inside pass A.attempt
    const bool trx_tab = A.trx_C != 0;                           // r is a position of the candidate table
    const ns_key key = make_key(A, trx_tab ? trx_key_rel(A, r) : r);
    const uint32_t a = meta_al ? A.attempt : trx_tab ? trx_attempt(A, r) : A.att_base[r] + A.attempt;
    const uint32_t epoch = (meta_al || trx_tab) ? 0u : A.rstate[r] & 0xffffu;
    uint32_t piece_off, n_pieces;
    if (trx_tab) { piece_off = (uint32_t)r; n_pieces = 1; }
    else if (A.attempt == 0 || meta_al) { piece_off = A.piece_off[r]; n_pieces = A.piece_off[r + 1] - piece_off; }
    else {                                           // a later pass: the read's slots, unless k_replan gave it new ones
        piece_off = A.reads[r].piece_off; n_pieces = A.reads[r].n_pieces;
        if (A.p_need && A.p_need[tid]) { n_pieces = A.p_need[tid]; piece_off = A.p_base + A.p_off[tid]; }
    }
    ns_piece *pc = A.pieces + piece_off;
    bool ok = true;
    uint64_t cap = 0, work = 0, work0 = 0;
    for (uint32_t pi = 0; pi < n_pieces; ++pi) {
        const bool is_gap = (kind == NS_KIND_UNALIGNED) || (pi & 1);
        int64_t mlen = 0;
        uint32_t plan_chrom = 0;
        if (prm.trx && kind != NS_KIND_UNALIGNED) {                                     // S:1082-1105: transcript and aligned length of
            const uint32_t pk = A.trx_cand[r];                                          // the candidate, planned by k_trx_walk
            if (pk == 0xffffffffu) ok = false;
            else {
                const uint64_t pi2 = (r / A.trx_C) * (uint64_t)A.trx_M + pk;
                plan_chrom = A.tx.expr_chrom[A.trx_pick_e[pi2]];
                mlen = A.trx_pick_y[pi2];
            }
        } else
        if (kind == NS_KIND_UNALIGNED) mlen = unaligned_length(A.m, prm, key, a);       // S:1494-1495
        else if (is_gap) mlen = gap_length(A.m, key, pi >> 1, A.meta ? a : epoch);     // S:1298-1299 (S:872: once per pass)
        else if (A.meta) mlen = A.m_len[A.m_segptr[r] + (pi >> 1)];                    // S:871: assigned by assign_species
        else if (!seg_length(A.m, prm, key, pi >> 1, epoch, mlen)) { ok = false; mlen = 0; }   // S:1285-1296
        const int32_t m32 = mlen > 0x3fffffff ? 0x3fffffff : mlen < -1 ? -1 : (int32_t)mlen;
        ns_piece p;
        cap = (cap + 3ull) & ~3ull;                                                     // every piece starts on a group of four event slots
        p.ref_gpos = 0; p.ev_off = cap /* RELATIVE until k_chain has run: its piece mode goes by it */; p.chrom = 0; p.pos = plan_chrom; p.ref_len = (uint32_t)m32; p.out_len = 0; p.n_ev = 0;   // (pos: the planned transcript until k_chain draws the start)
        p.kind = is_gap ? 1u : 0u;
        pc[pi] = p;
        const uint64_t l = m32 > 0 ? (uint64_t)m32 : 0;
        if (kind == NS_KIND_PERFECT) continue;
        if (is_gap) { cap += l * A.cap_gap_mul + 64; work += 20 * l; }    // a gap base costs ~20x an aligned base
        else { cap += (uint64_t)((double)l * A.cap_rate) + 64; work += l; }
        if (n_pieces > 1u) cap += 4;                                      // (k_chain starts every piece on a group of four events)
        if (pi == 0) work0 = work;
    }
    int32_t remainder = 0; double ratio = 0;
    if (prm.trx && kind == NS_KIND_ALIGNED) {                                          // S:1073-1076, 1203-1204: one draw per read, no filter
        const double x = ns_pow10m1(kde_sample(A.m.kde[NS_KDE_HT], ns_draw(key, ST_HT, 0, a, 0, 0)));
        const int64_t r64 = (int64_t)x;
        remainder = r64 < 0 ? 0 : r64 > 0x3fffffff ? 0x3fffffff : (int32_t)r64;
        ratio = kde_sample(A.m.kde[NS_KDE_RATIO], ns_draw(key, ST_RATIO, 0, a, 0, 0));
        if (ratio > 1) ratio = 1;
        if (ratio < 0) ratio = 0;
    } else
    if (kind == NS_KIND_ALIGNED && ok) {                                               // S:1471-1474, 1351-1352
        uint32_t j = 0;
        for (; j < NS_KDE_RETRY; ++j) {
            u32x4 w = ns_draw(key, ST_HT, 0, a, j, 0);
            double x = ns_pow10m1(kde_sample(A.m.kde[NS_KDE_HT], w));
            if (x >= 0) { remainder = A.meta ? (int32_t)rint(x) : (int32_t)x; break; }      // S:1351 int() / S:901 int(round())
        }
        for (j = 0; j < NS_KDE_RETRY; ++j) {
            u32x4 w = ns_draw(key, ST_RATIO, 0, a, j, 0);
            double x = kde_sample(A.m.kde[NS_KDE_RATIO], w);
            if (0 <= x && x <= 1) { ratio = x; break; }
        }
        if (j == NS_KDE_RETRY) ratio = 0.5;
    }
    u32x4 ws = ns_draw(key, ST_STRAND, 0, a, 0, 0);
    ns_read rd;
    rd.rec_off = 0; rd.piece_off = piece_off; rd.n_pieces = (uint16_t)n_pieces;
    rd.reversed = (u32_to_p(ws.x) > A.m.strandness_rate) ? 1 : 0;                       // S:1312, S:1524-1525
    if (meta_al) rd.reversed = (uint8_t)A.m_reversed;                                   // S:860: one draw per pass
    rd.flags = ok ? 1 : 3;                      // bit0: not generated yet, bit1: no valid length draw in this epoch
    rd.head = 0; rd.tail = 0;
    if (remainder != 0) {                                                               // S:1377-1382
        rd.head = (uint32_t)(int64_t)rint((double)remainder * ratio);
        rd.tail = (uint32_t)remainder - rd.head;
    }
    rd.seq_len = 0; rd.attempts = a;
    A.reads[r] = rd;
    cap = (cap + 3ull) & ~3ull;                  // whole 32-byte groups of events: k_chain flushes its staged events four at a time
    if (A.attempt > 0 && !meta_al && !trx_tab) A.l_cap[tid] = cap;
    if (A.attempt == 0 || meta_al || trx_tab) {
        A.ev_cap[r] = cap;
        // bit 31: a read of several pieces (chimeric) — those are visited first, together (visiting_order): a thread walks its pieces one
        // after the other, so ONE two-segment read among 64 made its whole wavefront run the second segment's trip count on top of the
        // first's (5 % chimeric reads: 96 % of the wavefronts; the chain of a chimeric batch took 4.6 instead of 2.0 ms).  Among themselves
        // they are ordered by (planned work, work of the FIRST piece), both on a coarse logarithmic scale (ord_coarse): the wavefront runs
        // max(first pieces) + max(the rest), and reads of equal total alone have those two anywhere.
        if (n_pieces > 1u) A.sort_key[r] = 0x80000000u | ord_coarse(work) << 5 | ord_coarse(work0);
        else A.sort_key[r] = work > 0x7fffffffull ? 0x7fffffffu : (uint32_t)work;
        A.sort_idx[r] = (uint32_t)r;
    }
}

// ---------------------------------------------------------------------------------------------------------
// k_chain: one thread per read: the error lists (S:1355-1365, S:1501), the acceptance tests (S:1367, S:1429),
// the start positions (S:1388-1389) and the record / error-log sizes.  Reads are visited in order of
// descending length (A.list) so that the 64 chains of a wavefront have similar trip counts; the chain tables
// live in LDS when they fit.  A rejected read is queued for the next pass (attempt a+1).
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// loop iterations per lane of the wave-per-read unaligned chain (coopk_unaligned_error_list): 4 = 116 VGPRs, 3 = 98, 2 = 82
#ifndef NS_UCOOP_ITER
#define NS_UCOOP_ITER 3
#endif
#ifndef NS_UCOOP_MINW
#define NS_UCOOP_MINW 4      // wavefronts per SIMD that kernel is compiled for
#endif
#define NS_STATS_WAYS 64u    // copies of the chain counters (k_chain -> k_stats_fold); == the threads of k_stats_fold
#define NS_STATS_BYTES ((8u + 8u * NS_STATS_WAYS) * sizeof(unsigned long long))
#ifndef NS_CHAIN_BLOCK
#define NS_CHAIN_BLOCK 256     // threads per block of the thread-per-read chain (320 was measured slower: 4.56 vs 4.09 ms)
#endif
// ... and when the LDS image is large (a trained model's hot prefixes: ~43 KB): 512 threads share one image — 43 + 16 KB of staging fit a
// CU twice = four waves per SIMD, where 256-thread workgroups (43 + 8 KB each) reach 2.6 (measured: profiles/r06/chain_trained_shape.log).
// (640 threads — ten waves, 3 + 3 + 2 + 2 over the SIMDs — would make five on paper, but two such workgroups rarely find their slots: 2.3.)
#ifndef NS_CHAIN_BLOCK_BIG
#define NS_CHAIN_BLOCK_BIG 512
#endif
#define NS_PIECE_SLOTS 6u          // k_chain's piece mode: threads per read of several pieces (five pieces = three segments one each; the sixth takes the rest)
// Wavefronts per SIMD the thread-per-read chain is compiled for.  Five: 95 VGPRs without scratch, and the bench model's LDS image (24.1 KB)
// + 8 KB of event staging fits five workgroups per CU (round 5, same box: aligned chain alone 2.94 -> 2.76 ms; four until round 4, when
// the lists needed 104 VGPRs and the image 30 KB)
#ifndef NS_CHAIN_MINW
#define NS_CHAIN_MINW 5
#endif
#ifdef NS_CHAIN_CLOCK
__device__ unsigned long long g_chain_clock[16];      // [multi ? 4 : 0] + {max, sum, waves, last block << 32 | its ticks}, [8..10] lane 0 of the multi wavefronts: aligned pieces, gaps, behind the lists: thread-per-read chain, per wavefront (100 MHz ticks; -DNS_CHAIN_CLOCK builds only)
#endif
template <bool LDS_TABLES, bool COOP, bool PIECES = false>       // PIECES: the piece modes (GenArgs.piece_mode) are compiled in
__global__ void __launch_bounds__(COOP ? 64 : NS_CHAIN_BLOCK_BIG, COOP ? (LDS_TABLES ? NS_UCOOP_MINW : 4) : NS_CHAIN_MINW) k_chain(GenArgs A) {
    extern __shared__ uint64_t lds_tbl[];
    CoopLds *coop = nullptr;
    if constexpr (COOP && !LDS_TABLES) { __shared__ CoopLds coop_lds; coop = &coop_lds; }
    Tabs T;
    if (LDS_TABLES) {
        // <true, true>: the wave-per-read UNALIGNED chain — all it reads is the front of the blob (n_words_mix words: the run-length
        // tables), 2-5 KB per wavefront.  From global memory the four dependent table reads of a block of 64 iterations were its
        // critical path (a 60 kb read: ~940 blocks of ~3 us set the duration of the batch's chain, 3.2 ms per 50 000 reads)
        const uint32_t nw = COOP ? A.m.ct.n_words_mix : A.m.ct.n_words_lds;
        for (uint32_t i = threadIdx.x; i < nw; i += blockDim.x) lds_tbl[i] = A.m.chain_blob[i];
        __syncthreads();
        T.w = lds_tbl;
    } else T.w = A.m.chain_blob;
    Tabs TM = T;                           // the front of the blob (transition rows, run-length tables)
    if constexpr (COOP && !LDS_TABLES) {
        // the wave-per-read ALIGNED chain: its future iterations' run lengths come from the front of the blob — three look-ups of three to four
        // dependent reads per block of 64 iterations: from an LDS copy when the launch made room for it (A.coop_mix: n_words_mix words)
        if (A.coop_mix) {
            for (uint32_t i = threadIdx.x; i < A.m.ct.n_words_mix; i += blockDim.x) lds_tbl[i] = A.m.chain_blob[i];
            __syncthreads();
            TM.w = lds_tbl;
        }
    }
    const ChainTab &ct = A.m.ct;
    // COOP: the whole wavefront works on ONE read (blockIdx.x-th entry of the list); lane 0 does the bookkeeping stores
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t tid = COOP ? (uint64_t)blockIdx.x : (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool lead = COOP ? lane == 0 : true;
    const ns_params &prm = A.prm;
    // The list is sorted by descending length, so the first workgroups carry the longest chains and set the makespan
    // (a 120 kb read is ~3800 dependent iterations): give them issue priority over the short-read waves they share a
    // SIMD with.
    // A batch with reads of several pieces is visited in TWO descending orders, those reads first (k_lengths): there the priority goes by
    // the planned work of the wavefront's reads against the work at ranks n/64, n/16, n/4 of the single-segment reads (prio_thr) —
    // by position, the reads of several pieces (5 % of the batch, 2-4 mean lengths each) held priorities 3 and 2 and the longest
    // single-segment reads ran at 1: their wavefronts took 2.0 ms instead of 0.5 (profiles/r06/ab_chimeric_order.log)
    if (!COOP && A.prio_thr) {
        uint32_t w = 0;
        if (tid < A.list_n) {
            const uint32_t k = A.sort_key[A.list[tid + (tid >= A.hole_at ? A.hole_len : 0u)]];
            w = (k >> 31) ? ord_coarse_floor((k >> 5) & 31u) : k;
        }
        for (int off = 32; off > 0; off >>= 1) w = max(w, (uint32_t)__shfl_xor((int)w, off));
        w = (uint32_t)__builtin_amdgcn_readfirstlane((int)w);
        if (w >= A.prio_thr[1]) __builtin_amdgcn_s_setprio(3);
        else if (w >= A.prio_thr[2]) __builtin_amdgcn_s_setprio(2);
        else if (w >= A.prio_thr[3]) __builtin_amdgcn_s_setprio(1);
    }
    else if (COOP || (PIECES && A.piece_mode == 1u) || blockIdx.x < (gridDim.x >> 6)) __builtin_amdgcn_s_setprio(3);
    else if (blockIdx.x < (gridDim.x >> 4)) __builtin_amdgcn_s_setprio(2);
    else if (blockIdx.x < (gridDim.x >> 2)) __builtin_amdgcn_s_setprio(1);
#ifdef NS_CHAIN_CLOCK
    const unsigned long long clk0 = wall_clock64(); uint32_t clk_multi = 0; unsigned long long clk_al = 0, clk_gap = 0, clk_pos = 0;
#endif
    unsigned long long st_over = 0, st_bases = 0, st_ref = 0, st_ev = 0;
    uint32_t st_max = 0;
    const bool piece_only = PIECES && !COOP && A.piece_mode == 1u;      // thread (slot, i): piece `slot` of the i-th read of the list
    const uint32_t slot = piece_only ? (uint32_t)(tid / A.list_n) : 0u;
    const uint64_t li = piece_only ? tid % A.list_n : tid;
    if (piece_only ? slot < NS_PIECE_SLOTS : tid < A.list_n) {
        const uint64_t r = A.list ? A.list[li + (li >= A.hole_at ? A.hole_len : 0u)] : A.list_base + li;      // (hole: the entries another launch takes)
        const int kind = (int)prm.kind;
        const bool trx_al = prm.trx && kind != NS_KIND_UNALIGNED;       // r is then a position of the candidate table (one try each)
        const bool meta_al = (A.meta && kind != NS_KIND_UNALIGNED) || trx_al;      // ... or of a metagenome pass
        const ns_key key = make_key(A, trx_al ? trx_key_rel(A, r) : r);
        const uint32_t a = trx_al ? trx_attempt(A, r) : meta_al ? A.attempt : A.att_base[r] + A.attempt;
        ns_read rd = A.reads[r];
        const uint32_t n_pieces = rd.n_pieces;
#ifdef NS_CHAIN_CLOCK
        clk_multi = n_pieces > 1u;
#endif
        ns_piece *pc = A.pieces + rd.piece_off;
        uint32_t epoch = meta_al ? 0u : A.rstate[r] & 0xffffu, fails = meta_al ? 0u : A.rstate[r] >> 16;
        bool accepted = false, overflow = false;
        do {
            if (rd.flags & 2) { ++epoch; fails = 0; break; }          // no valid length draw
            const uint64_t ev_off = A.l_off ? A.l_base + A.l_off[tid] : A.ev_base + A.ev_off[r];
            const uint64_t ev_cap64 = A.l_off ? A.l_off[tid + 1] - A.l_off[tid] : A.ev_off[r + 1] - A.ev_off[r];
            const uint32_t ev_cap = ev_cap64 > 0xffffffffull ? 0xffffffffu : (uint32_t)ev_cap64;
            EvSink32 sink; sink.last_ins_len = 0; sink.overflow = false; sink.range = false; sink.stride = blockDim.x;
            int64_t total = (int64_t)rd.head + rd.tail;
            uint32_t evn = 0, evp = 0;                               // events of the read so far / where the next piece's events start (below)
            const uint32_t trx_chrom = trx_al ? pc[0].pos : 0u;      // planned by k_lengths
            const uint32_t pi_lo = piece_only ? slot : 0u, pi_hi = (piece_only && slot + 1u < NS_PIECE_SLOTS) ? min(slot + 1u, n_pieces) : n_pieces;
            for (uint32_t pi = pi_lo; pi < pi_hi; ++pi) {
                ns_piece p = pc[pi];
                if (PIECES && !COOP && A.piece_mode == 2u) {                     // the list is done (piece mode 1): what it left in the record
                    const uint32_t fl = (uint32_t)(p.ref_gpos >> 32);
                    if (fl & 1u) sink.range = true;
                    if (fl & 2u) sink.overflow = true;
                    p.ev_off = ev_off + p.ev_off;
                    pc[pi] = p;
                    evn += p.n_ev;
                    if (!p.kind) total += (int64_t)(int32_t)(uint32_t)p.ref_gpos;            // e.l_new, S:1362
                    continue;
                }
                if (piece_only) {                                      // the piece's own share of the read's capacity (k_lengths)
                    evp = (uint32_t)p.ev_off;
                    const uint32_t end = pi + 1u < n_pieces ? (uint32_t)pc[pi + 1u].ev_off : ev_cap;
                    sink.range = false; sink.overflow = false;
                    sink.ev = A.events + ev_off + evp; sink.cap = end > evp ? end - evp : 0u;
                }
                const int32_t m32 = (int32_t)p.ref_len;                // planned length from k_lengths
                const uint32_t sid = p.kind ? NS_GAP_SEG + (pi >> 1) : (pi >> 1);
                if (!piece_only) { sink.ev = A.events + ev_off + evp; sink.cap = ev_cap > evp ? ev_cap - evp : 0; }
                sink.n = 0; sink.shift = 0;
                sink.stg = nullptr;
#ifdef NS_ABLATE
                if (COOP && LDS_TABLES && (A.dbg & (1u << 21))) sink.cap = 0;                       // (profiling: no event stores)
#endif
                if constexpr (LDS_TABLES && !COOP) {      // events leave in groups of four (32-byte stores), staged in LDS.  A read of several
                    // pieces starts every piece on a group boundary (up to three unused slots behind a piece: k_lengths plans them; every
                    // consumer goes by ns_piece.ev_off / n_ev).  Until round 6 such reads stored their events one by one, which made them
                    // ~2.2 times slower per base on top of being twice as long: the tail of a chimeric batch's chain launch (3.6 against 2.0 ms)
                    if (A.ev_stage) sink.stg = reinterpret_cast<uint2 *>(reinterpret_cast<uint8_t *>(lds_tbl) + A.ev_stage) + threadIdx.x;
                }
                EList32 e;
#ifdef NS_CHAIN_CLOCK
                const unsigned long long clk_p = wall_clock64();
#endif
#ifdef NS_ABLATE
                if (COOP && LDS_TABLES && (A.dbg & (1u << 20))) { e.l_new = e.middle_ref = m32; } else   // (profiling: no error list)
#endif
                if (kind == NS_KIND_PERFECT) { e.l_new = e.middle_ref = m32; }
                else if (p.kind) {
                    if constexpr (COOP && LDS_TABLES) e = A.coop_k1 ? coop_unaligned_error_list(TM, ct, m32, key, sid, a, sink, lane)
                                                                           : coopk_unaligned_error_list<NS_UCOOP_ITER>(TM, ct, m32, key, sid, a, sink, lane);
                    else e = COOP ? coop_unaligned_error_list(TM, ct, m32, key, sid, a, sink, lane) : chain_unaligned_error_list(T, ct, m32, key, sid, a, sink);
                }
                else if constexpr (COOP && LDS_TABLES) { e.l_new = e.middle_ref = m32; sink.range = true; }   // (not launched for aligned segments: their tables are not in this image)
                else if constexpr (COOP) e = coop_error_list(TM, T, ct, m32, key, sid, a, sink, *coop, lane);
                else if constexpr (LDS_TABLES) e = chain_error_list(T, Tabs{A.m.chain_blob}, ct, m32, key, sid, a, sink);
                else if (ct.int_image) e = chain_error_list(T, T, ct, m32, key, sid, a, sink);      // the integer image, from global memory (it does not fit LDS)
                else e = chain_error_list_g(T, ct, m32, key, sid, a, sink);
                ev_flush_tail(sink);
#ifdef NS_CHAIN_CLOCK
                if (p.kind) clk_gap += wall_clock64() - clk_p; else clk_al += wall_clock64() - clk_p;
#endif
#ifdef NS_ABLATE
                if (COOP && LDS_TABLES && (A.dbg & (1u << 21))) { sink.overflow = false; sink.n = 0; }
#endif
                if (!piece_only) p.ev_off = ev_off + evp;
                else p.ref_gpos = (uint64_t)(uint32_t)e.l_new | (sink.range ? 1ull << 32 : 0ull) | (sink.overflow ? 2ull << 32 : 0ull);
                p.ref_len = (uint32_t)(e.middle_ref < 0 ? 0 : e.middle_ref);
                p.out_len = (uint32_t)((e.middle_ref < 0 ? 0 : e.middle_ref) + sink.shift);
                p.n_ev = sink.n;
                p.chrom = (p.kind && kind == NS_KIND_ALIGNED && m32 == 0) ? 1u : 0u;   // empty gap marker (S:1553-1554)
                pc[pi] = p;          // COOP: every lane stores the same value (and later reads back its own store)
                evn += sink.n;
                evp += sink.stg ? (sink.n + 3u) & ~3u : sink.n;
                if (!p.kind) total += e.l_new;                                           // S:1362
                if (kind == NS_KIND_UNALIGNED) total = e.middle_ref;                     // S:1503
            }
            // An event that does not fit the 8-byte record (a run of more than NS_EV_LEN_MAX bases, a net insertion / deletion balance
            // beyond the 18-bit shift field: multi-megabase reads only): this ATTEMPT is dropped like one that fails the final length
            // check (S:1429-1430) and the read draws new lengths; counted in stats[0] >> 40 (ns_batch_info.n_range_redraws).  The
            // reference keeps Python integers (S:1875-1882) and would emit the read: a documented limit of the record format.
            if (piece_only) break;
            if (sink.range) { if (lead) st_over = 1ull << 40; if (!meta_al) { ++epoch; fails = 0; } break; }
            if (sink.overflow) { overflow = true; break; }
            int64_t trx_len = 0;
            if (trx_al) {                                    // S:1143-1144: middle_ref > ref_trx_len -> start over (no length limits)
                trx_len = (int64_t)(A.ref.chrom_off[trx_chrom + 1] - A.ref.chrom_off[trx_chrom]);
                if ((int64_t)pc[0].ref_len > trx_len) break;
            } else
            if (meta_al && !trx_al) {                        // S:907-946: remainder + middle_ref of the segments, then the gaps
                int64_t tot = (int64_t)rd.head + rd.tail; bool restart = false;
                for (uint32_t pi = 0; pi < n_pieces && !restart; pi += 2) { if (tot + pc[pi].ref_len > prm.max_len) restart = true; else tot += pc[pi].ref_len; }
                for (uint32_t pi = 1; pi < n_pieces && !restart; pi += 2) { if (tot + pc[pi].out_len > prm.max_len) restart = true; else tot += pc[pi].out_len; }
                if (restart || tot < prm.min_len || tot > prm.max_len) break;
            } else
            if (!trx_al && (total < prm.min_len || total > prm.max_len)) {               // S:1367-1368, S:1503-1504
                if (kind != NS_KIND_UNALIGNED && ++fails >= NS_EPOCH_FAILS) { ++epoch; fails = 0; }
                break;
            }
            if (A.defer_tail) { if (lead) A.accept[r] = 2ull | (uint64_t)evn << 32; break; }      // pending: k_meta_tail
            // ---- positions (S:1388-1389, 1510, 1557) ----
#ifdef NS_CHAIN_CLOCK
            clk_pos = wall_clock64();
#endif
            bool pos_ok = true;
            int64_t seq_len = (int64_t)rd.head + rd.tail;
            uint64_t ref_bases = 0;
            bool spliced = false, ir_reach = false;                  // intron retention (S:1156-1177)
            uint32_t ir_name = 0;
            for (uint32_t pi = 0; pi < n_pieces; ++pi) {
                ns_piece p = pc[pi];
                const uint32_t sid = p.kind ? NS_GAP_SEG + (pi >> 1) : (pi >> 1);
                uint32_t chrom = 0; uint64_t pos = 0;
                if (p.chrom == 1u && p.kind) { p.ref_len = 0; p.out_len = 0; p.n_ev = 0; }
                else if (trx_al) {
                    chrom = trx_chrom;
                    if (prm.model_ir && kind == NS_KIND_ALIGNED) {   // update_structure + extract_read_pos (S:1157-1160)
                        const IrPlan ip = ir_walk(A.ir, trx_chrom, p.ref_len, (uint32_t)trx_len, key, a,
                                                  [](uint32_t, uint32_t, uint32_t, uint32_t, bool) {});
                        if (ip.any) {
                            if (!ip.chrom_ok) { pos_ok = false; break; }                 // S:1167-1169
                            spliced = true; pos = ip.first_start;                        // S:1175
                            ir_reach = ip.last_end + 10u >= ip.struct_end;               // S:186
                            ir_name = ip.name_extra ? 16u + ip.name_extra : 0u;          // "_RetainedIntron_" + "<start>-<end>;" ... (S:1189-1192)
                        }
                    }
                    if (!spliced) {                                  // extract_read_trx (S:1683-1691): uniform start inside the transcript
                        const u32x4 wp = ns_draw(key, ST_POS, sid, a, 0, 0);
                        const uint64_t span = (uint64_t)(trx_len - (int64_t)p.ref_len) + 1;
                        pos = (uint64_t)(u53_to_p(wp.x, wp.y) * (double)span);
                        if (pos >= span) pos = span - 1;
                    }
                }
                else if (prm.trx) {                                  // unaligned read: any transcript longer than it (S:1695-1703)
                    if (!extract_pos_trx_any(A.ref, p.ref_len, key, sid, a, chrom, pos)) { pos_ok = false; break; }
                }
                else if (A.meta) {                                   // species of the segment; gaps / unaligned reads: any species
                    const int sp = (meta_al && !p.kind) ? (int)A.m_species[A.m_segptr[r] + (pi >> 1)] : -1;
                    if (!extract_pos_meta(A.ref, A.species_chrom_off, A.nspecies, p.ref_len, sp, key, sid, a, chrom, pos)) { pos_ok = false; break; }
                }
                else if (!extract_pos(A.ref, p.ref_len, key, sid, a, chrom, pos)) { pos_ok = false; break; }
                p.chrom = chrom; p.pos = (uint32_t)pos;
                p.ref_gpos = spliced ? NS_SPLICED_BASE : A.ref.chrom_off[chrom] + pos;   // (spliced: k_ir_splice fills in the arena offset)
                pc[pi] = p;
                seq_len += p.out_len;
                ref_bases += p.ref_len;
            }
            uint32_t polya = 0;                                      // S:1046-1053, 1206-1209: exponential tail if the read reaches the 3' end
            if (trx_al && pos_ok && A.tx.polya && A.tx.polya[trx_chrom] &&
                (spliced ? ir_reach : (int64_t)pc[0].pos + (int64_t)pc[0].ref_len + 10 >= trx_len)) {
                const u32x4 wa = ns_draw(key, ST_TRX, 0, a, 1, 0);
                const int64_t pl = (int64_t)fma(A.tx.polya_scale, -ns_log(u32_to_p(wa.x)), 2.0);    // int(expon.rvs(loc=2, scale))
                polya = pl > 65535 ? 65535u : (uint32_t)pl;
                seq_len += polya;
            }
            // final length re-check (S:1429-1430, S:1518-1519); with -k the length is only final after k_hp_drain
            if (!pos_ok || (!A.hp && !trx_al && (seq_len < prm.min_len || seq_len > prm.max_len))) { ++epoch; fails = 0; break; }
            // ---- accepted ----
            rd.flags = 0; rd.seq_len = (uint32_t)seq_len; rd.attempts = a;
            uint32_t nl = 0; bool first = true;                                          // name length (S:1390-1402, 1332-1343, 1529-1534)
            for (uint32_t pi = 0; pi < n_pieces; ++pi) {
                ns_piece p = pc[pi];
                if (p.kind && kind == NS_KIND_ALIGNED) {
                    if (A.meta) nl += 5 + dec_digits(p.out_len);                        // ";gap_<len>" (S:970-971)
                    continue;
                }
                if (!first) nl += 2;            // ';' in the position list and ';' in the length list
                first = false;
                nl += (A.ref.name_off[p.chrom + 1] - A.ref.name_off[p.chrom] - 1) + 1 + dec_digits(p.pos) + dec_digits(p.ref_len);
            }
            // metagenome: the number of the read is only known once the accepted reads of the pass are counted (k_meta_commit)
            nl += (kind == NS_KIND_UNALIGNED ? 11u : 9u) + (meta_al ? 0u : dec_digits(A.name_first + r)) + ir_name;
            if (kind == NS_KIND_ALIGNED && n_pieces > 1) nl += 9;
            nl += 2 /*_F*/ + 1 + dec_digits(rd.head) + 1 + 1 + dec_digits(rd.tail + polya);       // S:1211-1213: tail + polya_len
            uint64_t err_len = 0;
            if (prm.emit_errlog && !A.errlen_later) {
                for (uint32_t pi = 0; pi < n_pieces; ++pi) {
                    ns_piece p = pc[pi];
                    if (p.kind) continue;
                    const ns_event *ev = A.events + p.ev_off;
                    for (uint32_t j = 0; j < p.n_ev; ++j) {
                        ns_event e = ev[j];
                        err_len += nl + dec_digits(e.pos) + dec_digits(ns_ev_len(e.info)) + 2u * ns_ev_len(e.info) + 9u;
                    }
                }
            }
            if (lead) {
                A.name_len[r] = (uint16_t)nl;
                A.rec_len[r] = prm.emit_records ? (uint64_t)nl + 2 + (uint64_t)seq_len + 1 + (prm.fastq ? (uint64_t)seq_len + 3 : 0) : 0;
                A.err_len[r] = err_len;
                if (A.polya) A.polya[r] = (uint16_t)polya;
                if (A.ir_need) A.ir_need[r] = spliced ? ir_slot_bytes(pc[0].ref_len) : 0;
                if (meta_al) { A.accept[r] = 1ull | (uint64_t)n_pieces << 32; A.sort_key[r] = evn; }   // (event count: taken back if -k rejects the read)
                st_bases = A.hp ? 0ull : (unsigned long long)seq_len; st_ref = ref_bases; st_ev = evn; st_max = (uint32_t)seq_len;
                if (trx_al) st_bases = st_ref = st_ev = 0;       // (a candidate is not a read yet: k_trx_commit counts those it takes)
            }
            accepted = true;
        } while (false);
        if (lead && !piece_only) {
            if (overflow) st_over += 1;
            A.reads[r] = rd;
            if (meta_al) { /* a rejected read is re-planned by the next pass */ }
            else if (accepted) A.att_base[r] = a;       // a re-run of the batch starts every read at its accepted attempt
            if (!meta_al && !accepted && !overflow) {
                A.rstate[r] = (epoch & 0xffffu) | fails << 16;
                A.next_list[atomicAdd(A.next_n, 1u)] = (uint32_t)r;
            }
        }
    }
#ifdef NS_CHAIN_CLOCK
    if (!COOP && A.attempt == 0) {
        const unsigned long long dt = wall_clock64() - clk0;
        const bool any_multi = __ballot(clk_multi != 0) != 0;
        if ((threadIdx.x & 63) == 0) { unsigned long long *C = g_chain_clock + (any_multi ? 4 : 0); atomicMax(&C[0], dt); atomicAdd(&C[1], dt); atomicAdd(&C[2], 1ull);
                                       atomicMax(&C[3], (unsigned long long)blockIdx.x << 32 | dt);
                                       if (any_multi) { atomicAdd(&g_chain_clock[8], clk_al); atomicAdd(&g_chain_clock[9], clk_gap); atomicAdd(&g_chain_clock[10], clk_pos ? wall_clock64() - clk_pos : 0ull); } }
    }
#endif
    // one atomic per wavefront and counter
    st_over = wave_sum(st_over); st_bases = wave_sum(st_bases); st_ref = wave_sum(st_ref); st_ev = wave_sum(st_ev);
    if (A.prm.kind == NS_KIND_UNALIGNED) for (int off = 32; off > 0; off >>= 1) st_max = max(st_max, (uint32_t)__shfl_xor((int)st_max, off));
    // ... into one of NS_STATS_WAYS copies of the counters (k_stats_fold adds them up behind the chain kernels).  On ONE set of counters the
    // 200 000 atomics of a wave-per-read launch over 50 000 unaligned reads were executed one after the other at the memory side, ~12 ns
    // each: they, not the error lists, were the 2.6 ms of that kernel (round 6: the kernel without its lists took 2.44 ms)
    if ((threadIdx.x & 63) == 0) {
        unsigned long long *S = A.stats + 8u + 8u * (blockIdx.x & (NS_STATS_WAYS - 1u));
        if (st_over) atomicAdd(&S[0], st_over);
        if (st_bases) atomicAdd(&S[1], st_bases);
        if (st_ref) atomicAdd(&S[2], st_ref);
        if (st_ev) atomicAdd(&S[3], st_ev);
        if (A.prm.kind == NS_KIND_UNALIGNED && st_max) atomicMax(&S[4], (unsigned long long)st_max);
    }
}

// the NS_STATS_WAYS copies of the chain counters -> stats[0..4]; the copies are left zeroed for the next launch
// zero_me: the counter of the record kernel's slow-tile queue (a 4-byte hipMemsetAsync in front of that kernel is a launch of its own that waits
// for room next to the other call's kernels: 0.1 ms in the timeline of a step)
__global__ void __launch_bounds__(64) k_stats_fold(unsigned long long *stats, uint32_t *zero_me) {
    if (threadIdx.x == 0 && zero_me) *zero_me = 0;
    unsigned long long *S = stats + 8u + 8u * threadIdx.x;
    unsigned long long v[5];
    #pragma unroll
    for (int k = 0; k < 5; ++k) { v[k] = S[k]; S[k] = 0; }
    #pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = wave_sum(v[k]);
    for (int off = 32; off > 0; off >>= 1) { const unsigned long long o = __shfl_xor(v[4], off); v[4] = o > v[4] ? o : v[4]; }
    if (threadIdx.x == 0) {
        #pragma unroll
        for (int k = 0; k < 4; ++k) if (v[k]) atomicAdd(&stats[k], v[k]);
        if (v[4]) atomicMax(&stats[4], v[4]);
    }
}

// ---------------------------------------------------------------------------------------------------------
// The visiting order of a batch: reads by descending planned work, so that the 64 chains of a wavefront have similar trip counts and the
// longest reads come first.  Nothing depends on the order beyond that (a read is a function of its index), so it need not be a sort: the
// reads are dealt into 1 024 bins of their key on a logarithmic scale (32 bins per octave: keys of a bin differ by at most 3 %), bins in
// descending order, reads inside a bin in whatever order they arrive — twice: the reads of several pieces (chimeric) first, then the rest.  Three small kernels instead of rocPRIM's merge sort of 10^6 pairs
// (a block sort + 15 merge passes, ~0.16 of the 0.28 ms of a call's planning phase).
// ---------------------------------------------------------------------------------------------------------
#define NS_ORD_BINS 2048u                                     // [0, 1024): reads of several pieces (key bit 31), [1024, 2048): the others
__device__ __forceinline__ uint32_t ord_bin(uint32_t key31) {
    if (key31 >> 31) return 1023u - (key31 & 1023u);         // several pieces: k_lengths made the two coarse bins, descending
    const uint32_t key = key31;
    uint32_t v = key;                                        // keys below 32: one bin each
    if (key >= 32u) { const uint32_t e = 31u - (uint32_t)__clz((int)key); v = (e - 4u) * 32u + ((key >> (e - 5u)) & 31u); }   // <= 863
    return 1024u + 1023u - v;                                // descending
}
__global__ void __launch_bounds__(1024) k_order_hist(const uint32_t *__restrict__ keys, uint32_t n, uint32_t *__restrict__ hist) {
    __shared__ uint32_t h[NS_ORD_BINS];
    h[threadIdx.x] = 0; h[threadIdx.x + 1024u] = 0;
    __syncthreads();
    for (uint32_t i = blockIdx.x * 1024u + threadIdx.x; i < n; i += gridDim.x * 1024u) atomicAdd(&h[ord_bin(keys[i])], 1u);
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
    if (h[threadIdx.x + 1024u]) atomicAdd(&hist[threadIdx.x + 1024u], h[threadIdx.x + 1024u]);
}
// exclusive scan of the bins -> the cursor of every bin; cursor[NS_ORD_BINS] = the reads of several pieces; the histogram is left zeroed
__global__ void __launch_bounds__(1024) k_order_scan(uint32_t *__restrict__ hist, uint32_t *__restrict__ cursor) {
    __shared__ uint32_t wsum[16];
    const uint32_t c0 = hist[2u * threadIdx.x], c1 = hist[2u * threadIdx.x + 1u];     // thread t: bins 2t, 2t + 1
    hist[2u * threadIdx.x] = 0; hist[2u * threadIdx.x + 1u] = 0;
    const uint32_t c = c0 + c1;
    const uint32_t incl = wave_incl_scan(c);
    if ((threadIdx.x & 63u) == 63u) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) base += wsum[w];
    base += incl - c;
    cursor[2u * threadIdx.x] = base; cursor[2u * threadIdx.x + 1u] = base + c0;
    if (threadIdx.x == 512u) cursor[NS_ORD_BINS] = base;     // (bin 1024 starts here)
    // cursor[NS_ORD_BINS + 1 .. 3]: the work of the single-segment read at ranks n/64, n/16, n/4 of their order (k_chain's issue priorities)
    __shared__ uint32_t n_multi_s, n_all_s;
    if (threadIdx.x == 512u) n_multi_s = base;
    if (threadIdx.x == 1023u) n_all_s = base + c;
    __syncthreads();
    if (threadIdx.x >= 512u && c) {
        const uint32_t n_single = n_all_s - n_multi_s, lo = base - n_multi_s;       // ranks [lo, lo + c) of the single-segment reads
#pragma unroll
        for (uint32_t j = 0; j < 3; ++j) {
            const uint32_t rank = n_single >> (6u - 2u * j);
            if (lo <= rank && rank < lo + c) {
                const uint32_t bin = 2u * threadIdx.x + (rank < lo + c0 ? 0u : 1u), v = 2047u - bin;     // ord_bin backwards: the smallest key of the bin
                cursor[NS_ORD_BINS + 1u + j] = v < 32u ? v : (32u + (v & 31u)) << ((v >> 5) - 1u);
            }
        }
    }
}
// a workgroup deals 1 024 consecutive reads: ranks inside the workgroup from LDS counters, ONE global atomic per bin the workgroup touches
__global__ void __launch_bounds__(1024) k_order_deal(const uint32_t *__restrict__ keys, uint32_t n, uint32_t *__restrict__ cursor, uint32_t *__restrict__ list) {
    __shared__ uint32_t cnt[NS_ORD_BINS];
    cnt[threadIdx.x] = 0; cnt[threadIdx.x + 1024u] = 0;
    __syncthreads();
    const uint32_t r = blockIdx.x * 1024u + threadIdx.x;
    uint32_t b = 0, rank = 0;
    if (r < n) { b = ord_bin(keys[r]); rank = atomicAdd(&cnt[b], 1u); }
    __syncthreads();
    const uint32_t m0 = cnt[threadIdx.x], m1 = cnt[threadIdx.x + 1024u];
    __syncthreads();
    if (m0) cnt[threadIdx.x] = atomicAdd(&cursor[threadIdx.x], m0);                   // (now: where this workgroup's reads of the bin start)
    if (m1) cnt[threadIdx.x + 1024u] = atomicAdd(&cursor[threadIdx.x + 1024u], m1);
    __syncthreads();
    if (r < n) list[cnt[b] + rank] = r;
}

// ---------------------------------------------------------------------------------------------------------
// k_meta_tail: what k_chain does behind the error lists, for the reads of a metagenome pass it left pending (A.defer_tail): the start
// positions in the species assign_species gave the segments (extract_read, S:1704-1749), the final length check (S:1023-1024), the name
// and record sizes, the pass's acceptance flag.  Thread per pass position.  The pass launches its lists BEFORE the host has walked the
// quotas (the lists do not depend on the species): the walk — sequential by definition, ~3 ms per 10^6 reads — then runs next to the
// chain kernels instead of in front of them.  (The statements are those of k_chain's metagenome branch; change both.)
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_meta_tail(GenArgs A) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const ns_params &prm = A.prm;
    unsigned long long st_bases = 0, st_ref = 0, st_ev = 0;
    const uint64_t pend = r < A.list_n ? A.accept[r] : 0ull;
    if ((pend & 3ull) == 2ull) {
        const uint32_t evn = (uint32_t)(pend >> 32), a = A.attempt;
        const ns_key key = make_key(A, r);
        ns_read rd = A.reads[r];
        const uint32_t n_pieces = rd.n_pieces;
        ns_piece *pc = A.pieces + rd.piece_off;
        bool pos_ok = true;
        int64_t seq_len = (int64_t)rd.head + rd.tail;
        uint64_t ref_bases = 0;
        for (uint32_t pi = 0; pi < n_pieces; ++pi) {
            ns_piece p = pc[pi];
            const uint32_t sid = p.kind ? NS_GAP_SEG + (pi >> 1) : (pi >> 1);
            uint32_t chrom = 0; uint64_t pos = 0;
            if (p.chrom == 1u && p.kind) { p.ref_len = 0; p.out_len = 0; p.n_ev = 0; }
            else {                                           // species of the segment; gaps: any species
                const int sp = !p.kind ? (int)A.m_species[A.m_segptr[r] + (pi >> 1)] : -1;
                if (!extract_pos_meta(A.ref, A.species_chrom_off, A.nspecies, p.ref_len, sp, key, sid, a, chrom, pos)) { pos_ok = false; break; }
            }
            p.chrom = chrom; p.pos = (uint32_t)pos;
            p.ref_gpos = A.ref.chrom_off[chrom] + pos;
            pc[pi] = p;
            seq_len += p.out_len;
            ref_bases += p.ref_len;
        }
        uint64_t acc = 0;
        if (pos_ok && (A.hp || (seq_len >= prm.min_len && seq_len <= prm.max_len))) {      // S:1023-1024; with -k the length is only final after k_hp_drain
            rd.flags = 0; rd.seq_len = (uint32_t)seq_len; rd.attempts = a;
            uint32_t nl = 0; bool first = true;                                              // name length (S:965-985)
            for (uint32_t pi = 0; pi < n_pieces; ++pi) {
                const ns_piece p = pc[pi];
                if (p.kind && prm.kind == NS_KIND_ALIGNED) { nl += 5 + dec_digits(p.out_len); continue; }      // ";gap_<len>" (S:970-971)
                if (!first) nl += 2;
                first = false;
                nl += (A.ref.name_off[p.chrom + 1] - A.ref.name_off[p.chrom] - 1) + 1 + dec_digits(p.pos) + dec_digits(p.ref_len);
            }
            nl += 9u;                                        // (the number of the read: k_meta_commit, once the accepted reads of the pass are counted)
            if (prm.kind == NS_KIND_ALIGNED && n_pieces > 1) nl += 9;
            nl += 2 /*_F*/ + 1 + dec_digits(rd.head) + 1 + 1 + dec_digits(rd.tail);
            A.name_len[r] = (uint16_t)nl;
            A.rec_len[r] = prm.emit_records ? (uint64_t)nl + 2 + (uint64_t)seq_len + 1 + (prm.fastq ? (uint64_t)seq_len + 3 : 0) : 0;
            A.err_len[r] = 0;                                // (k_errlen / k_hp_filter_w)
            A.sort_key[r] = evn;                             // (event count: taken back if -k rejects the read)
            acc = 1ull | (uint64_t)n_pieces << 32;
            st_bases = A.hp ? 0ull : (unsigned long long)seq_len; st_ref = ref_bases; st_ev = evn;
        }
        A.accept[r] = acc;
        A.reads[r] = rd;
    }
    st_bases = wave_sum(st_bases); st_ref = wave_sum(st_ref); st_ev = wave_sum(st_ev);
    if ((threadIdx.x & 63) == 0 && (st_bases | st_ref | st_ev)) {
        unsigned long long *S = A.stats + 8u + 8u * (blockIdx.x & (NS_STATS_WAYS - 1u));
        atomicAdd(&S[1], st_bases); atomicAdd(&S[2], st_ref); atomicAdd(&S[3], st_ev);
    }
}

// ---------------------------------------------------------------------------------------------------------
// metagenome passes (simulation_aligned_metagenome, S:836-1036)
//   k_meta_draw    the length list of a pass: one KDE draw per remaining segment (S:852), keyed by the batch
//   k_meta_commit  numbers the accepted reads of a pass consecutively (S:909-911) and moves them to their final slots
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_meta_draw(GenArgs A) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= A.draw_n) return;
    const ns_key key = make_key(A, 0);
    const uint32_t jl = (uint32_t)j, jh = (uint32_t)(j >> 32) << 1;          // sub = 2 * high part (+ 1 for the second block of a draw)
    const u32x4 w = ns_draw(key, ST_REFLEN, 0, A.attempt, jl, jh);
    double x;
    if (!A.prm.use_lognormal) x = kde_sample(A.m.kde[NS_KDE_ALIGNED], w);                                   // S:852
    else if (A.prm.kind == NS_KIND_PERFECT)                                                                   // S:840
        x = ns_exp(fma(A.prm.sd_len, ns_norminv(u32_to_p(w.z)), ns_log(A.prm.median_len)));
    else {                                                                                                    // S:854-856: total - remainder
        const double tot = ns_exp(fma(A.prm.sd_len, ns_norminv(u32_to_p(w.z)), ns_log(A.prm.median_len + A.prm.sd_len * A.prm.sd_len / 2)));
        const double rem = ns_pow10m1(kde_sample(A.m.kde[NS_KDE_HT], ns_draw(key, ST_REFLEN, 0, A.attempt, jl, jh | 1u)));
        x = rem < 0 ? -1.0 : tot - rem;                                        // (a negative remainder is filtered out, S:846)
    }
    A.draw_x[j] = x;
}

// the random.choice / random.uniform words of assign_species for every segment pointer of the pass (S:786-803), keyed by the batch
__global__ void __launch_bounds__(256) k_meta_words(GenArgs A, uint2 *out, uint64_t n) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const u32x4 w = ns_draw(make_key(A, 0), ST_SPECIES, 0, A.attempt, (uint32_t)j, (uint32_t)(j >> 32));
    out[j] = make_uint2(w.x, w.y);
}
// int(round(length)) of the assigned lengths (S:871), on the device copy of the sorted list
__global__ void __launch_bounds__(256) k_meta_round(const double *__restrict__ x, int32_t *__restrict__ out, uint64_t n) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) out[j] = (int32_t)rint(x[j]);
}
struct MetaLenFilter {           // S:857 (0 < x <= max_l) / S:841 (--perfect: min_l <= x <= max_l)
    double lo, hi; bool lo_inclusive;
    __host__ __device__ bool operator()(const double &x) const { return (lo_inclusive ? lo <= x : lo < x) && x <= hi; }
};

// dst[key] += val for every lane with key != ~0u: one atomic per distinct key and wavefront (the species quotas are ten addresses;
// one atomic per read on them cost 4 ms per 10^6 reads)
__device__ inline void wave_add_by_key(unsigned long long *dst, uint32_t key, unsigned long long val) {
    for (;;) {
        const uint64_t act = __ballot(key != 0xffffffffu);
        if (!act) break;
        const int first = __builtin_ctzll(act);
        const uint32_t lead = (uint32_t)__shfl((int)key, first);
        const bool mine = key == lead;
        const unsigned long long sum = wave_sum(mine ? val : 0ull);
        if ((int)(threadIdx.x & 63u) == first) atomicAdd(&dst[lead], sum);
        if (mine) key = 0xffffffffu;
    }
}

__global__ void __launch_bounds__(256) k_meta_commit(GenArgs A) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool on = i < A.list_n && A.accept[i];
    ns_read rd; rd.n_pieces = 0; rd.piece_off = 0;
    uint64_t sc = 0;
    if (on) { rd = A.reads[i]; sc = A.accept_scan[i]; }
    const uint32_t poff = A.m_pieces_passed + (uint32_t)(sc >> 32);
    const ns_piece *src = A.pieces + rd.piece_off;
    ns_piece *dst = A.f_pieces + poff;
    uint64_t rows = 0;
    uint32_t maxp = on ? rd.n_pieces : 0u;
    for (int off = 32; off > 0; off >>= 1) maxp = max(maxp, (uint32_t)__shfl_xor((int)maxp, off));
    for (uint32_t pi = 0; pi < maxp; ++pi) {
        uint32_t sp = 0xffffffffu; unsigned long long bases = 0;
        if (on && pi < rd.n_pieces) {
            const ns_piece p = src[pi];
            dst[pi] = p;
            if (!p.kind) {
                rows += p.n_ev;
                if (A.prm.kind == NS_KIND_ALIGNED) {                   // S:1001-1002 (the --perfect branch never updates the quotas)
                    sp = A.m_species[A.m_segptr[i] + (pi >> 1)]; bases = p.ref_len;
                }
            }
        }
        // (one of NS_STATS_WAYS copies of the species counters, summed by the host: on one copy the ~150 000 atomics of a 10^6-read pass —
        // one per wavefront and species — were executed one after the other: 1.1 of this kernel's 1.2 ms, profiles/r06/ab_meta_commit.log)
        wave_add_by_key(A.species_bases + (size_t)(blockIdx.x & (NS_STATS_WAYS - 1u)) * A.nspecies, sp, bases);
    }
    if (!on) return;
    const uint64_t slot = A.m_passed + (uint32_t)sc;
    rd.piece_off = poff;
    A.f_reads[slot] = rd;
    A.key_pos_w[slot] = (uint32_t)i;
    const uint32_t dg = dec_digits(A.name_first + slot);
    A.f_name_len[slot] = (uint16_t)(A.name_len[i] + dg);
    A.f_rec_len[slot] = A.prm.emit_records ? A.rec_len[i] + dg : 0;
    A.f_err_len[slot] = A.prm.emit_errlog ? A.err_len[i] + rows * dg : 0;
}

// ---------------------------------------------------------------------------------------------------------
// transcriptome, aligned / --perfect batches: which transcript, which aligned length (S:1080-1104)
// The reference worker keeps ONE sample of the 2-D KDE until a transcript it has used is picked again (trx_sampled, S:1087-1092); inside
// a sample a transcript always gets the same aligned length (the nearest point, S:108-111), so one whose length failed S:1103-1104 keeps
// failing until some other transcript repeats.  Restated per BLOCK of NS_TRX_BLOCK read indices (whatever batch or rank generates
// them) as a walk over the block's own sequence of picks — the oracle's trx_block:
//   k_trx_picks   pick j of a block: transcript by expression (random.choices, S:1084) and the aligned length it gets under a FRESH
//                 sample (a draw from the KDE conditioned on the transcript length: kde2d_cond) — thread per pick
//   (radix sort)  by (block, transcript, pick): the previous pick of the same transcript
//   k_trx_walk    the sequential part, one wavefront per block: pick j is inside the current sample if its transcript was picked since
//                 the sample started; it starts a new sample if that earlier pick succeeded, fails like it if it failed, and is
//                 evaluated with its own draw otherwise; a successful pick is the block's next CANDIDATE
//   k_lengths / k_chain over the candidate table (one try per candidate, key = (block start + c mod W, attempt c / W))
//   k_trx_commit  the reads of a block = its first NS_TRX_BLOCK surviving candidates; those of the batch move to their slots
// ---------------------------------------------------------------------------------------------------------
#define NS_TRX_PICK_BITS 20u       // picks per block (< 2^20) and transcripts (< 2^22) in the sort key: block[21] | transcript[22] | pick[20]
__global__ void __launch_bounds__(256) k_trx_picks(GenArgs A, uint64_t n_picks, uint32_t M, uint64_t block0, uint32_t *__restrict__ pick_e,
                                                   int32_t *__restrict__ pick_y, uint64_t *__restrict__ keys) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_picks) return;
    const uint64_t blk = i / M; const uint32_t j = (uint32_t)(i % M);
    const uint64_t g = (block0 + blk) * NS_TRX_BLOCK;
    const ns_key key{(uint32_t)A.prm.seed, (uint32_t)(A.prm.seed >> 32), (uint32_t)g, (uint32_t)(g >> 32)};
    const u32x4 w = ns_draw(key, ST_TRX, 0, 0, j, 2);
    const uint32_t e = trx_pick(A.tx, u53_to_p(w.x, w.y));                                // S:1084
    const uint32_t chrom = A.tx.expr_chrom[e];
    const int64_t L = (int64_t)(A.ref.chrom_off[chrom + 1] - A.ref.chrom_off[chrom]);
    const int64_t y = kde2d_cond(A.m, (double)L, key, 0, 1u + j);                          // S:1098-1102
    pick_e[i] = e;
    pick_y[i] = (y > 0 && y < L) ? (int32_t)y : -1;                                        // S:1103-1104
    keys[i] = blk << (22u + NS_TRX_PICK_BITS) | (uint64_t)e << NS_TRX_PICK_BITS | j;
}
// sorted keys -> the previous pick of the same transcript in the same block (-1: none)
__global__ void __launch_bounds__(256) k_trx_prev(const uint64_t *__restrict__ keys, uint64_t n_picks, uint32_t M, int32_t *__restrict__ prev) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_picks) return;
    const uint64_t k = keys[i], blk = k >> (22u + NS_TRX_PICK_BITS);
    const uint32_t j = (uint32_t)k & ((1u << NS_TRX_PICK_BITS) - 1u);
    int32_t p = -1;
    if (i) { const uint64_t q = keys[i - 1]; if ((q >> NS_TRX_PICK_BITS) == (k >> NS_TRX_PICK_BITS)) p = (int32_t)((uint32_t)q & ((1u << NS_TRX_PICK_BITS) - 1u)); }
    prev[blk * M + j] = p;
}
// the walk: one wavefront per block; the state is the first pick of the current sample and, per pick, whether its transcript's length
// holds under the sample it belongs to (ok[], one byte per pick in LDS); 64 picks are fetched at a time and handed over by readlane
__global__ void __launch_bounds__(64) k_trx_walk(uint32_t M, uint32_t C, const int32_t *__restrict__ prev, const int32_t *__restrict__ pick_y,
                                                 uint32_t *__restrict__ cand, unsigned long long *__restrict__ n_short) {
    extern __shared__ uint8_t ok_lds[];
    volatile uint8_t *ok = ok_lds;
    const uint32_t lane = threadIdx.x;
    const uint64_t blk = blockIdx.x;
    const int32_t *pv = prev + blk * M, *py = pick_y + blk * M;
    uint32_t *out = cand + blk * C;
    uint32_t c = 0;
    int32_t start = 0;                                     // first pick of the current sample
    for (uint32_t j0 = 0; j0 < M && c < C; j0 += 64) {
        const int32_t p_l = j0 + lane < M ? pv[j0 + lane] : -1, y_l = j0 + lane < M ? py[j0 + lane] : -1;
        const uint32_t kn = min(64u, M - j0);
        for (uint32_t k = 0; k < kn && c < C; ++k) {
            const int32_t p = __builtin_amdgcn_readlane(p_l, (int)k), y = __builtin_amdgcn_readlane(y_l, (int)k);
            const int32_t j = (int32_t)(j0 + k);
            bool inside = p >= start;                      // the transcript was picked before under this sample
            if (inside && ok[p]) { start = j; inside = false; }        // ... and used: a new sample starts with this pick (S:1087-1092)
            const bool good = !inside && y >= 0;           // (inside: it failed before, the same nearest point fails again)
            if (lane == 0) ok[j] = good ? 1 : 0;
            if (good) { if (lane == 0) out[c] = (uint32_t)j; ++c; }
            wave_sync();
        }
    }
    for (uint32_t q = c + lane; q < C; q += 64) out[q] = 0xffffffffu;
    if (lane == 0 && c < C) atomicAdd(n_short, 1ull);      // the picks ran out before the table was full: the host asks for more picks
}
// the candidates that survived (k_chain: accept) -> the reads of their blocks, those of the batch [g0, g0 + n) -> their final slots
__global__ void __launch_bounds__(256) k_trx_commit(GenArgs A, uint64_t n_pos, uint64_t block0, uint64_t g0, uint64_t n, uint16_t *__restrict__ polya_out,
                                                    uint64_t *__restrict__ ir_need_out, unsigned long long *__restrict__ n_short) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long st_bases = 0, st_ref = 0, st_ev = 0;
    if (i < n_pos) {
        const uint64_t blk = i / A.trx_C, c = i % A.trx_C;
        const uint64_t acc = A.accept[i] & 1ull, before = (A.accept_scan[i] & 0xffffffffull) - (A.accept_scan[blk * A.trx_C] & 0xffffffffull);
        const uint64_t gb = (block0 + blk) * NS_TRX_BLOCK, g = gb + before;
        if (acc && before < NS_TRX_BLOCK && g >= g0 && g < g0 + n) {
            const uint64_t slot = g - g0;
            ns_read rd = A.reads[i];
            ns_piece p = A.pieces[rd.piece_off];
            rd.piece_off = (uint32_t)slot;
            A.f_reads[slot] = rd; A.f_pieces[slot] = p;
            A.key_pos_w[slot] = (uint32_t)(blk * NS_TRX_BLOCK + c % NS_TRX_BLOCK);
            const uint32_t dg = dec_digits(A.name_first + slot);
            A.f_name_len[slot] = (uint16_t)(A.name_len[i] + dg);
            A.f_rec_len[slot] = A.prm.emit_records ? A.rec_len[i] + dg : 0;
            A.f_err_len[slot] = 0;                              // (k_errlen / k_hp_filter_w size the rows of the final reads)
            if (polya_out) polya_out[slot] = A.polya[i];
            if (ir_need_out) ir_need_out[slot] = A.ir_need[i];
            st_bases = A.hp ? 0ull : rd.seq_len; st_ref = p.ref_len; st_ev = p.n_ev;
        }
        if (c == A.trx_C - 1u) {                                // the block's census: enough survivors for the reads the batch needs of it?
            const uint64_t total = before + acc, end = g0 + n;
            const uint64_t need = gb >= end ? 0ull : min((uint64_t)NS_TRX_BLOCK, end - gb);
            if (total < need) atomicAdd(n_short, 1ull);
        }
    }
    st_bases = wave_sum(st_bases); st_ref = wave_sum(st_ref); st_ev = wave_sum(st_ev);
    if ((threadIdx.x & 63) == 0 && (st_bases | st_ref | st_ev)) {         // (one of the copies of the counters: k_stats_fold — on one set these
        unsigned long long *S = A.stats + 8u + 8u * (blockIdx.x & (NS_STATS_WAYS - 1u));      // 94 000 atomics were 1.1 of the kernel's 1.2 ms)
        atomicAdd(&S[1], st_bases); atomicAdd(&S[2], st_ref); atomicAdd(&S[3], st_ev);
    }
}

// ---------------------------------------------------------------------------------------------------------
// k_names: record framing and read name, one thread per read
// ---------------------------------------------------------------------------------------------------------
// The header line of a read (">name\n") is composed by its thread in a row of LDS and leaves in contiguous stores — lane = byte, one read of
// the wavefront after the other.  Written by the composing thread straight into the record image (until round 6) it was ~50 single-byte
// stores per read, every store instruction of a wavefront touching 64 different cache lines of an image the record kernel is filling at the
// same time on the other stream: the step was 0.2-0.3 ms longer with k_names than without it (profiles/r06/ab_names.log).  A header
// longer than a row (chimeric reads with many segments) is written the old way.
#define NS_NAME_ROW 128u
__global__ void __launch_bounds__(256) k_names(GenArgs A) {
    __shared__ uint8_t rows[256][NS_NAME_ROW];
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    uint64_t rec_off = 0;
    uint32_t hdr = 0;                        // bytes of the header in this thread's row (0: nothing to copy)
    if (r < A.prm.n_reads) {
        ns_read rd = A.reads[r];
        rd.rec_off = A.rec_off[r];
        A.reads[r].rec_off = rd.rec_off;
        rec_off = rd.rec_off;
        if (!rd.flags && A.prm.emit_records == 1u) {
            const int kind = (int)A.prm.kind;
            const ns_piece *pc = A.pieces + rd.piece_off;
            const uint32_t name_len = A.name_len[r];                   // without '>' and '\n' (k_chain / k_meta_commit)
            const bool in_row = name_len + 2u <= NS_NAME_ROW;
            // (a lambda inlined at two call sites: behind each the compiler knows the address space of `p` — LDS stores for the row, global
            // stores for the image; through one pointer that may be either they were flat stores, several times slower)
            auto compose = [&](uint8_t *p) -> uint8_t * {
                *p++ = A.prm.fastq ? '@' : '>';
                bool first = true;
                for (uint32_t pi = 0; pi < rd.n_pieces; ++pi) {
                    if (pc[pi].kind && kind == NS_KIND_ALIGNED) {
                        if (A.meta) { const char *g = ";gap_"; while (*g) *p++ = (uint8_t)*g++; p = put_dec(p, pc[pi].out_len); }   // S:970-971
                        continue;
                    }
                    if (!first) *p++ = ';';
                    first = false;
                    const char *cn = A.ref.names + A.ref.name_off[pc[pi].chrom];
                    while (*cn) *p++ = (uint8_t)*cn++;
                    *p++ = '_';
                    p = put_dec(p, pc[pi].pos);
                }
                const char *tag = kind == NS_KIND_ALIGNED ? "_aligned_" : kind == NS_KIND_PERFECT ? "_perfect_" : "_unaligned_";
                while (*tag) *p++ = (uint8_t)*tag++;
                p = put_dec(p, A.name_first + r);
                if (kind == NS_KIND_ALIGNED && rd.n_pieces > 1) { const char *c = "_chimeric"; while (*c) *p++ = (uint8_t)*c++; }
                if (pc[0].ref_gpos >= NS_SPLICED_BASE) {                          // "_RetainedIntron_<start>-<end>;..." (S:1189-1192)
                    const uint32_t trx = pc[0].chrom;
                    const uint32_t trx_len = (uint32_t)(A.ref.chrom_off[trx + 1] - A.ref.chrom_off[trx]);
                    bool open = false;
                    ir_walk(A.ir, trx, pc[0].ref_len, trx_len, read_key(A, r), rd.attempts, [&](uint32_t, uint32_t, uint32_t start, uint32_t end, bool retained) {
                        if (!retained) return;
                        if (!open) { const char *c = "_RetainedIntron_"; while (*c) *p++ = (uint8_t)*c++; open = true; }
                        p = put_dec(p, start); *p++ = '-'; p = put_dec(p, end); *p++ = ';';
                    });
                }
                *p++ = '_'; *p++ = rd.reversed ? 'R' : 'F';
                *p++ = '_'; p = put_dec(p, rd.head);
                *p++ = '_';
                first = true;
                for (uint32_t pi = 0; pi < rd.n_pieces; ++pi) {
                    if (pc[pi].kind && kind == NS_KIND_ALIGNED) continue;
                    if (!first) *p++ = ';';
                    first = false;
                    p = put_dec(p, pc[pi].ref_len);
                }
                *p++ = '_'; p = put_dec(p, rd.tail + (A.polya ? A.polya[r] : 0u));
                *p++ = '\n';
                return p;
            };
            uint32_t len;
            if (in_row) { uint8_t *row = rows[threadIdx.x]; len = (uint32_t)(compose(row) - row); hdr = len; }
            else { uint8_t *img = A.records + rd.rec_off; len = (uint32_t)(compose(img) - img); }
            // the framing behind the bases (and the qualities): single bytes far from the header
            uint8_t *g = A.records + rd.rec_off + len + rd.seq_len;
            *g++ = '\n';
            if (A.prm.fastq) { *g++ = '+'; *g++ = '\n'; g += rd.seq_len; *g++ = '\n'; }
        }
    }
    // copy-out: the 64 rows of this wavefront, one after the other, lane = byte
    __syncthreads();                         // (the rows were written through generic pointers by other lanes: wait for them)
    const uint32_t row0 = threadIdx.x & ~63u;
    uint64_t todo = __ballot(hdr != 0);
    while (todo) {
        const int j = __builtin_ctzll(todo);
        todo &= todo - 1;
        const uint32_t n = (uint32_t)__shfl((int)hdr, j);
        const uint64_t off = (uint64_t)(uint32_t)__shfl((int)(uint32_t)rec_off, j) | (uint64_t)(uint32_t)__shfl((int)(uint32_t)(rec_off >> 32), j) << 32;
        uint8_t *dst = A.records + off;
        const uint8_t *src = rows[row0 + (uint32_t)j];
        for (uint32_t b = lane; b < n; b += 64u) dst[b] = src[b];
    }
}

// ---------------------------------------------------------------------------------------------------------
// k_ir_splice: intron retention — one wavefront per read whose structure has a retained intron: the exons / retained introns under
// the read are copied from the genome into the read's slot of the splice arena (S:1161-1178), in the orientation of the transcript
// (reverse_complement for strand '-', which only knows upper-case ACGT, S:1675-1680) and in the device form of the bases
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64 * NS_WPB) k_ir_splice(GenArgs A) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t r = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= A.prm.n_reads) return;
    const uint64_t off = A.ir.arena_off[r];
    if (A.ir.arena_off[r + 1] == off) return;
    const ns_read rd = A.reads[r];
    if (rd.flags) return;
    ns_piece *pp = A.pieces + rd.piece_off;
    const ns_key key = read_key(A, r);
    const uint32_t trx = pp->chrom, L = pp->ref_len;
    const uint32_t trx_len = (uint32_t)(A.ref.chrom_off[trx + 1] - A.ref.chrom_off[trx]);
    uint8_t *dst = A.ir.arena + off + NS_IR_PAD;
    const IrPlan ip = ir_walk(A.ir, trx, L, trx_len, key, rd.attempts, [](uint32_t, uint32_t, uint32_t, uint32_t, bool) {});
    uint32_t done = 0;
    ir_walk(A.ir, trx, L, trx_len, key, rd.attempts, [&](uint32_t, uint32_t chrom, uint32_t start, uint32_t end, bool) {
        const uint8_t *src = A.ir.genome + A.ir.genome_off[chrom] + start;
        for (uint32_t i = lane; i < end - start; i += 64) {
            uint32_t c = src[i];
            if (ip.minus) {
                c = c == 'A' ? 'T' : c == 'T' ? 'A' : c == 'C' ? 'G' : c == 'G' ? 'C' : c;
                dst[L - 1 - (done + i)] = normalise_base(c);
            } else dst[done + i] = normalise_base(c);
        }
        done += end - start;
    });
    if (lane == 0) pp->ref_gpos = NS_SPLICED_BASE + off + NS_IR_PAD;
}

// ---------------------------------------------------------------------------------------------------------
// k_materialise: one read per wavefront (64-thread workgroup); see ns_materialise.h
// ---------------------------------------------------------------------------------------------------------
#ifndef NS_MAT_WAVES
#define NS_MAT_WAVES 7
#endif

// read header of a wave-per-read kernel: everything wave-uniform, pinned to SGPRs
__device__ __forceinline__ bool load_read_uniform(const GenArgs &A, uint64_t r, bool fastq, ns_read &rd, ns_key &key, ReadOut &ro) {
    rd = A.reads[r];
    if (rd.flags) return false;
    rd.rec_off = uni64(A.rec_off[r]);        // (the scan's value: k_names, which files it in the read, may still be running)
    rd.piece_off = uni(rd.piece_off); rd.n_pieces = (uint16_t)uni(rd.n_pieces);
    rd.reversed = (uint8_t)uni(rd.reversed); rd.head = uni(rd.head); rd.tail = uni(rd.tail); rd.seq_len = uni(rd.seq_len);
    rd.attempts = uni(rd.attempts);
    key = read_key(A, r);
    key.r_lo = uni(key.r_lo); key.r_hi = uni(key.r_hi);
    ro.seq = A.records + rd.rec_off + uni(A.name_len[r]) + 2;
    ro.qual = fastq ? ro.seq + rd.seq_len + 3 : nullptr;
    ro.seq_len = rd.seq_len; ro.reversed = rd.reversed != 0; ro.uracil = A.prm.uracil != 0;
    return true;
}

// k_materialise: the sequence line of one read per wavefront; see ns_materialise.h.  k_qualities: its quality line, NS_MATQ_WAVES reads
// per workgroup, which share the LDS copy of the quality bucket tables (reads of similar length: `order` is the batch's length-sorted
// visiting list, so no wavefront idles long behind its workgroup's longest read).
#ifndef NS_MATQ_WAVES
#define NS_MATQ_WAVES 4
#endif
#ifndef NS_MATQ_MINW
#define NS_MATQ_MINW 6          // waves per SIMD k_qualities is compiled for (80 VGPRs; for 8 it spills: 11.25 against 11.0 ms per FASTQ batch)
#endif
// first event slot / capacity of the homopolymer edits of the piece whose scratch bytes start at `pos` (bytes from the start of the
// scratch buffer) and that is the `piece`-th piece of the batch IN READ ORDER (A.hp_pord[read] + its number in the read — the order of
// the scratch buffer): a monotone function of both, so the slots of no two pieces overlap
__device__ __forceinline__ uint64_t hp_ev_slot(const GenArgs &A, uint64_t pos, uint32_t piece) { return (pos >> A.hp_shift) + (uint64_t)A.hp_pad * piece; }

// chunks per lane and tile: two (2 KB tiles) where a tile's ~63 events span about that much; the second pass of -k has an event every
// ~300 bytes, so its tiles are byte-limited and a larger one amortises the per-tile work — by little: 4.78 ms with 2 KB tiles, 4.66 with
// 3 KB, 4.70 with 4 KB at seven waves (8 bytes of scratch), 4.60 with 4 KB at six waves, 4.71 with 6 KB at four (same box, under
// rocprofv3: profiles/r06/ab_final_pass_tiles.log).  What the pass costs is per chunk, not per tile.
#ifndef NS_TILE_CHUNKS_FINAL
#define NS_TILE_CHUNKS_FINAL 4u
#endif
#ifndef NS_MAT_WAVES_FINAL
#define NS_MAT_WAVES_FINAL 6
#endif
template <bool FASTQ, int MODE>
__global__ void __launch_bounds__(64, MODE == MAT_HP_FINAL ? NS_MAT_WAVES_FINAL : NS_MAT_WAVES)
k_materialise(GenArgs A, const uint32_t *ev_word, uint32_t dbg, SlowQueue sq, const uint32_t *order) {
    constexpr bool CLSOUT = FASTQ && MODE != MAT_HP_SCRATCH;           // FASTQ: the bases here, the quality line in k_qualities
    constexpr uint32_t TC = MODE == MAT_HP_FINAL ? NS_TILE_CHUNKS_FINAL : NS_TILE_CHUNKS;
    __shared__ TileLds7<TC> T;
    const uint32_t lane = threadIdx.x;
    const uint64_t slot = blockIdx.x;
    if (slot >= A.prm.n_reads) return;
    const uint64_t r = order ? (uint64_t)uni(order[slot]) : slot;
    ns_read rd; ns_key key; ReadOut ro;
    if (!load_read_uniform(A, r, false, rd, key, ro)) return;
    const uint32_t a = rd.attempts;
    tile_lds_init(T, lane);
    uint32_t *cls = nullptr;
    if constexpr (CLSOUT) cls = A.cls + cls_word0(rd.rec_off, r, cls_per_read(A.prm));
    if constexpr (MODE == MAT_HP_SCRATCH) {
        // -k, first pass: the pieces of the read before mutate_homo, forward strand, one after the other in the scratch buffer
        // (head, tail and polyA are written by the second pass; strand and T -> U are applied there)
        ro.seq = A.scr + uni64(A.scr_off[r]); ro.qual = nullptr; ro.reversed = false; ro.uracil = false;
        uint32_t q = 0;
        for (uint32_t pi = 0; pi < rd.n_pieces; ++pi) {
            const PieceCtx pc = load_piece_uniform(A.events, A.ref, A.pieces[rd.piece_off + pi], pi, ev_word);
            materialise_piece7<FASTQ, MODE, TC>(A.m, A.ref, T, ro, key, a, pc, q, lane, dbg, sq, (uint32_t)r, pi, nullptr);
            q += pc.out_len;
        }
        return;
    }
#ifdef NS_ABLATE
    if (!(dbg & 8))
#endif
    emit_head_tail(A.m, ro, key, a, rd.head, rd.tail, lane);                                                              // S:1426-1427
    uint32_t q = rd.head;
    uint64_t q_in = MODE == MAT_HP_FINAL ? uni64(A.scr_off[r]) : 0;
    for (uint32_t pi = 0; pi < rd.n_pieces; ++pi) {
        PieceCtx pc;
        if constexpr (MODE == MAT_HP_FINAL) {
            // the source is the scratch piece, the events are the homopolymer edits k_hp_drain filed for it (mutate_homo, S:618-705)
            const uint32_t gp = rd.piece_off + pi;
            const ns_piece p = A.pieces[gp];
            const uint64_t eo = hp_ev_slot(A, q_in, uni(A.hp_pord[r]) + pi);
            pc.kind = uni(p.kind);
            pc.ev = A.hp_ev + eo; pc.wd = A.hp_wd + eo; pc.n_ev = pc.kind ? 0u : uni(A.hp_nev[gp]);
            pc.ref_len = uni(p.out_len); pc.out_len = pc.kind ? pc.ref_len : uni(A.hp_len[gp]);
            pc.chrom_base = (uint64_t)((uintptr_t)A.scr - (uintptr_t)A.ref.bases) + q_in; pc.chrom_len = ~0ull; pc.pos = 0;
            pc.sid = pc.kind ? NS_GAP_SEG + (pi >> 1) : (pi >> 1);
            q_in += pc.ref_len;
        } else pc = load_piece_uniform(A.events, A.ref, A.pieces[rd.piece_off + pi], pi, ev_word);
        materialise_piece7<FASTQ, MODE, TC>(A.m, A.ref, T, ro, key, a, pc, q, lane, dbg, sq, (uint32_t)r, pi, CLSOUT ? cls + (q >> 4) + 2u * pi : nullptr);
        q += pc.out_len;                                 // (-k: k_hp_report files the emitted length in the piece once the record kernels are done)
    }
    if (A.polya) {                                                                          // transcriptome: polyA tail (S:1224-1225)
        const uint32_t pl = uni(A.polya[r]);
        if (pl) emit_polya(A.m, ro, key, a, q, pl, rd.head, rd.tail, lane);
    }
}

// k_qualities: the quality line of one read per wavefront (predict_base_qualities per class, bq:120-130, the truncated log-normal of bq:9-20; classes S:1421-1423, 1953-1955,
// 1564), from the class words k_materialise<true, .> left.  HPF: second record pass of -k (piece lengths after mutate_homo).
template <bool HPF>
__global__ void __launch_bounds__(64 * NS_MATQ_WAVES, NS_MATQ_MINW) k_qualities(GenArgs A, const uint32_t *order) {
    __shared__ __align__(16) uint16_t qlut[NS_QLUT_SLOTS * 1024u];
    qual_lut_load(qlut, A.m, threadIdx.x, 64 * NS_MATQ_WAVES);
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t slot = (uint64_t)blockIdx.x * NS_MATQ_WAVES + (threadIdx.x >> 6);
    if (slot >= A.prm.n_reads) return;
    const uint64_t r = order ? (uint64_t)uni(order[slot]) : slot;
    ns_read rd; ns_key key; ReadOut ro;
    if (!load_read_uniform(A, r, true, rd, key, ro)) return;
    const uint32_t a = rd.attempts;
    QualState Q; Q.lut = qlut;
    qualities_head_tail(A.m, Q, ro, key, a, rd.head, rd.tail, lane);                         // S:1421-1423
    const uint32_t *cls = A.cls + cls_word0(rd.rec_off, r, cls_per_read(A.prm));
    uint32_t q = rd.head;
    for (uint32_t pi = 0; pi < rd.n_pieces; ++pi) {
        const uint32_t gp = rd.piece_off + pi;
        const ns_piece p = A.pieces[gp];
        const uint32_t kind = uni(p.kind);
        const uint32_t out_len = (HPF && !kind) ? uni(A.hp_len[gp]) : uni(p.out_len);
        qualities_piece(A.m, Q, ro, key, a, kind ? NS_GAP_SEG + (pi >> 1) : (pi >> 1), kind, out_len, q, cls + (q >> 4) + 2u * pi, lane);
        q += out_len;
    }
    if (A.polya) {
        const uint32_t pl = uni(A.polya[r]);
        if (pl) emit_polya_quals(A.m, ro, key, a, q, pl, rd.head, rd.tail, lane);
    }
}

// Unaligned reads (S:1482-1549): ~0.55 events per base.  The tiled kernel ends a tile after 63 events — every ~110 bases here, twenty
// prologues per read — so these reads take the lane-per-event path (ns_materialise.h: dense_piece).
#define NS_DENSE_SEG 4096u        // output bytes of a read one wavefront writes: the longest read of a batch (tens of kb) no longer sets the kernel's duration
// work list of the dense kernel: stretches of NS_DENSE_SEG output bytes per read (at least one: an empty read still has its framing), then
// an exclusive scan — workgroup w finds its (read, stretch) by binary search.  (A grid of n x ceil(longest read / NS_DENSE_SEG) made
// nine workgroups in ten exit at once on a 2 kb batch, and one 1 Mb read in a batch multiplied the launch by 245.)
__global__ void __launch_bounds__(256) k_dense_plan(GenArgs A, uint32_t *cnt) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > A.prm.n_reads) return;
    uint32_t c = 0;
    if (r < A.prm.n_reads) { const ns_read rd = A.reads[r]; c = rd.flags ? 0u : max(1u, (rd.seq_len + NS_DENSE_SEG - 1u) / NS_DENSE_SEG); }
    cnt[r] = c;
}
// (FASTQ = true draws the qualities in place through the per-value look-up; launch_materialise uses <false> + k_qualities since round 3)
template <bool FASTQ>
__global__ void __launch_bounds__(64) k_materialise_dense(GenArgs A, const uint32_t *__restrict__ seg_off) {
    __shared__ DenseLds S;
    const uint32_t lane = threadIdx.x;
    const uint32_t w = blockIdx.x, n = (uint32_t)A.prm.n_reads;
    if (w >= seg_off[n]) return;
    uint32_t lo = 0, hi = n;                                   // the read whose stretches hold w: last r with seg_off[r] <= w
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (seg_off[mid] <= w) lo = mid; else hi = mid; }
    const uint64_t r = uni(lo);
    const uint32_t seg = uni(w - seg_off[lo]);
    ns_read rd; ns_key key; ReadOut ro;
    if (!load_read_uniform(A, r, FASTQ, rd, key, ro)) return;
    const uint32_t a = rd.attempts;
    if (seg == 0) emit_head_tail(A.m, ro, key, a, rd.head, rd.tail, lane);
    uint32_t q = rd.head;
    for (uint32_t pi = 0; pi < rd.n_pieces; ++pi) {
        const PieceCtx pc = load_piece_uniform(A.events, A.ref, A.pieces[rd.piece_off + pi], pi);
        // bytes of the piece that fall into this wavefront's stretch [seg, seg + 1) * NS_DENSE_SEG of the read (head excluded)
        const uint64_t s_lo = (uint64_t)seg * NS_DENSE_SEG, s_hi = s_lo + NS_DENSE_SEG, p_lo = q - rd.head, p_hi = p_lo + pc.out_len;
        if (p_hi > s_lo && p_lo < s_hi) {
            const uint32_t m_lo = (uint32_t)(max(s_lo, p_lo) - p_lo), m_hi = (uint32_t)(min(s_hi, p_hi) - p_lo);
            if (A.dbg & 16384u) slow_piece_range(A.m, A.ref, ro, key, a, pc, q, m_lo, m_hi, lane);      // (profiling: the per-byte path)
            else dense_piece<FASTQ>(A.m, A.ref, S, ro, key, a, pc, q, m_lo, m_hi, lane);
        }
        q += pc.out_len;
    }
}

// the tiles k_materialise could not take: generic per-byte path, one wavefront per queued tile.  SCRATCH: first record pass of -k
template <bool FASTQ, bool SCRATCH>
__global__ void __launch_bounds__(64) k_materialise_slow(GenArgs A, SlowQueue sq) {
    const uint32_t lane = threadIdx.x;
    const uint32_t n = min(*sq.count, sq.cap);
    for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
        const SlowTile t = sq.items[i];
        ns_read rd; ns_key key; ReadOut ro;
        if (!load_read_uniform(A, t.read, FASTQ && !SCRATCH, rd, key, ro)) continue;
        uint32_t q = SCRATCH ? 0u : rd.head;
        if constexpr (SCRATCH) { ro.seq = A.scr + A.scr_off[t.read]; ro.qual = nullptr; ro.reversed = false; ro.uracil = false; }
        for (uint32_t pi = 0; pi < t.piece; ++pi) q += A.pieces[rd.piece_off + pi].out_len;
        const PieceCtx pc = load_piece_uniform(A.events, A.ref, A.pieces[rd.piece_off + t.piece], t.piece);
        slow_piece_range<FASTQ && SCRATCH>(A.m, A.ref, ro, key, rd.attempts, pc, q, t.m0, t.m1, lane);
    }
}

// the tiles of the SECOND record pass of -k that k_materialise<., MAT_HP_FINAL> could not take (>= 64 homopolymer edits at one output
// offset: a stretch of adjacent runs all re-sampled to nothing): generic per-byte path over the scratch piece and its edit list, one
// wavefront per queued tile.  (Until round 2 such a tile failed the batch with NS_EINVAL: the condition depends on the reference — a
// low-complexity stretch — not on the caller.)
template <bool FASTQ>
__global__ void __launch_bounds__(64) k_materialise_slow_hpf(GenArgs A, SlowQueue sq) {
    const uint32_t lane = threadIdx.x;
    const uint32_t n = min(*sq.count, sq.cap);
    for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
        const SlowTile t = sq.items[i];
        ns_read rd; ns_key key; ReadOut ro;
        if (!load_read_uniform(A, t.read, FASTQ, rd, key, ro)) continue;
        uint32_t q = rd.head;                                    // output offset of the piece in the read / of its scratch bytes
        uint64_t q_in = A.scr_off[t.read];
        for (uint32_t pi = 0; pi < t.piece; ++pi) {
            const uint32_t gp = rd.piece_off + pi;
            const ns_piece p = A.pieces[gp];
            q += p.kind ? p.out_len : A.hp_len[gp];
            q_in += p.out_len;
        }
        const uint32_t gp = rd.piece_off + t.piece;
        const ns_piece p = A.pieces[gp];
        const uint64_t eo = hp_ev_slot(A, q_in, A.hp_pord[t.read] + t.piece);
        const ns_event *ev = A.hp_ev + eo;
        const uint32_t *wd = A.hp_wd + eo;
        const uint32_t n_ev = p.kind ? 0u : A.hp_nev[gp], sid = p.kind ? NS_GAP_SEG + (t.piece >> 1) : (t.piece >> 1);
        const uint8_t *src = A.scr + q_in;
        for (uint32_t m0 = t.m0 + lane * 16; m0 < t.m1; m0 += 64 * 16) {
            const uint32_t count = min(16u, t.m1 - m0);
            uint32_t lo = 0, hi = n_ev;                          // events with out_start <= m0
            while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (ev_out_start(ev[mid]) <= m0) lo = mid + 1; else hi = mid; }
            uint32_t j = lo;                                     // event in force: j - 1 (none: copy from the start of the piece)
            uint64_t blo = 0, bhi = 0, qlo = 0, qhi = 0;
            QualDraw qd; qd.blk = 0xffffffffu;
            for (uint32_t i2 = 0; i2 < count; ++i2) {
                const uint32_t m = m0 + i2;
                while (j < n_ev && ev_out_start(ev[j]) <= m) ++j;
                uint32_t b, cls_bits;
                if (j == 0) { const uint32_t c = src[m]; b = c & ~(NS_CLS_MIS_BIT | NS_CLS_INS_BIT); cls_bits = c & (NS_CLS_MIS_BIT | NS_CLS_INS_BIT); }
                else {
                    const ns_event e = ev[j - 1];
                    const uint32_t len = ns_ev_len(e.info), ty = ns_ev_type(e.info), os = ev_out_start(e), d = m - os, w = wd[j - 1];
                    const uint32_t pl = ty == NS_DEL ? 0u : len;
                    if (d < pl) {
                        if (ty == NS_INS) {                      // inserted letters: 2-bit fields of the edit's word; letter 0 may be the run's first mismatch
                            b = bases_atcg((w >> (2 * (d & 15))) & 3u);
                            cls_bits = (d == 0 && (w >> 31)) ? NS_CLS_MIS_BIT : NS_CLS_INS_BIT;
                        } else {                                 // one substituted base: the first base-3 digit of the word picks among the other three
                            const uint32_t c = src[e.pos + d];
                            uint32_t f = w;
                            b = mis_from_digit(c & ~(NS_CLS_MIS_BIT | NS_CLS_INS_BIT), next_digit3(f));
                            cls_bits = (w & 1u) ? NS_CLS_MIS_BIT : (c & (NS_CLS_MIS_BIT | NS_CLS_INS_BIT));
                        }
                    } else {
                        const uint32_t c = src[e.pos + (ty == NS_INS ? 0u : len) + (d - pl)];
                        b = c & ~(NS_CLS_MIS_BIT | NS_CLS_INS_BIT); cls_bits = c & (NS_CLS_MIS_BIT | NS_CLS_INS_BIT);
                    }
                }
                put_byte(blo, bhi, i2, b);
                if (ro.qual) {
                    const int cls = p.kind ? NS_Q_UNMAPPED : (cls_bits & NS_CLS_MIS_BIT) ? NS_Q_MIS : (cls_bits & NS_CLS_INS_BIT) ? NS_Q_INS : NS_Q_MATCH;
                    put_byte(qlo, qhi, i2, qual_draw(qd, A.m, cls, key, ST_QUAL, sid, rd.attempts, m));
                }
            }
            store_chunk(ro, q + m0, count, blo, bhi, qlo, qhi);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// -k kernels (ns_hp.h)
// ---------------------------------------------------------------------------------------------------------
// Homopolymer bitmap of the reference for one k, two bits per base (position p: bits 2p and 2p + 1 of the byte array):
//   bit 0: p lies in a run of >= k identical unambiguous bases of the reference as stored (runs are not cut at chromosome ends: a
//          segment never crosses one, and a run a segment cuts is looked at again — k_hp_filter_w);
//   bit 1: an IUPAC code within k - 1 bases of p: what case_convert (S:743-755) makes of it is the read's own draw, so the runs around
//          it are, and the filter walks the bases.
// Built once per (reference, k); the filter then tests an event with one 8-byte load instead of a 16-byte window of the reference
// per k bases of the event (13.2 KB of reference per 8.4 kb read, 3.6 ms per 950 000 reads).
__global__ void __launch_bounds__(256) k_hp_bitmap(const uint8_t *__restrict__ bases, uint64_t n, uint32_t k, uint8_t *__restrict__ bm) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t * 4 >= n) return;
    uint32_t out = 0;
    for (uint32_t i = 0; i < 4; ++i) {
        const uint64_t p = t * 4 + i;
        if (p >= n) break;
        const uint8_t b = bases[p];
        bool amb = (b & 0x80u) != 0;
        uint32_t l = 0, r = 0;
        bool lrun = !amb, rrun = !amb;
        for (uint32_t j = 1; j < k; ++j) {
            if (p >= j) { const uint8_t c = bases[p - j]; if (c & 0x80u) amb = true; if (lrun && c == b) ++l; else lrun = false; } else lrun = false;
            if (p + j < n) { const uint8_t c = bases[p + j]; if (c & 0x80u) amb = true; if (rrun && c == b) ++r; else rrun = false; } else rrun = false;
        }
        if (!(b & 0x80u) && l + r + 1 >= k) out |= 1u << (2 * i);
        if (amb) out |= 2u << (2 * i);
    }
    bm[t] = (uint8_t)out;
}

// -k filter of mutate_read (S:1929-1947), one read per wavefront: lane per event (the homopolymer test of an event is independent of the others), ballot /
// prefix-popcount compaction, exclusive wavefront prefix sum of the length changes for the shift field
// (80 VGPRs = six waves per SIMD as it stands.  Round 6, profiles/r06/ab_hp_filter.log: asking for the next block's events and bitmap windows one
// iteration ahead — the kernel waits for two dependent loads per 64 events — needs 89 VGPRs = five waves and bought 0.08 ms of 2.4; bounding the
// kernel to six waves explicitly moves 12 bytes to scratch: 2.41 -> 2.60 ms.  Both gone again.)
__global__ void __launch_bounds__(64 * NS_WPB) k_hp_filter_w(GenArgs A) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t r = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r > A.prm.n_reads) return;
    if (r == A.prm.n_reads) { if (lane == 0) { A.scr_len[r] = 0; A.hp_pcnt[r] = 0; } return; }
    ns_read rd = A.reads[r];
    if (rd.flags) { if (lane == 0) { A.scr_len[r] = 0; A.hp_pcnt[r] = 0; } return; }
    if (lane == 0) A.hp_pcnt[r] = rd.n_pieces;
    const ns_key key = read_key(A, r);
    const uint32_t a = rd.attempts;
    const int64_t k = (int64_t)A.prm.kmer_bias;
    const uint32_t nl = A.name_len[r];
    uint64_t seq_len = (uint64_t)rd.head + rd.tail, err_len = 0, scr_len = 0;
    for (uint32_t pi = 0; pi < rd.n_pieces; ++pi) {
        ns_piece p = A.pieces[rd.piece_off + pi];
        if (!p.kind) {
            const PieceCtx pc = load_piece(A.events, A.ref, p, pi);
            ns_event *ev = A.events + p.ev_off;
            uint32_t w = 0; int32_t shift = 0;                            // kept events / their cumulative length change so far
            const bool win = k <= 16 && !(pc.pos + pc.ref_len > pc.chrom_len);   // (a segment across the origin takes the generic walk)
            const bool win8 = win && k <= 8;
            const uint8_t *bm = (win && pc.chrom_len != ~0ull) ? A.hp_bm : nullptr;   // (not for spliced stretches: they lie in the arena)
            for (uint32_t j0 = 0; j0 < p.n_ev; j0 += 64) {                // S:1929-1947
                const uint32_t j = j0 + lane;
                const bool valid = j < p.n_ev;
                ns_event e; e.pos = 0; e.info = 0;
                if (valid) e = ev[j];
                const int64_t pos = e.pos, len = ns_ev_len(e.info); const uint32_t ty = ns_ev_type(e.info);
                bool keep = valid, walk = valid;
                if (valid && bm) {                  // the bitmap decides unless an IUPAC code is near, or the run it shows may be cut by the segment's ends
                    const int64_t lo = max((int64_t)(ty == NS_INS ? pos - 1 : pos), (int64_t)0), hi = min(pos + len - 1, (int64_t)pc.ref_len - 1);
                    walk = false;
                    for (int64_t x = lo; x <= hi; x += 28) {
                        const uint64_t gb = 2ull * (pc.chrom_base + pc.pos + (uint64_t)x);
                        uint64_t v;
                        __builtin_memcpy(&v, bm + (gb >> 3), 8);
                        v >>= (gb & 7u);
                        const uint32_t nb = (uint32_t)min((int64_t)28, hi - x + 1);
                        v &= (1ull << (2u * nb)) - 1ull;
                        if (v & 0xaaaaaaaaaaaaaaaaull) { walk = true; break; }
                        if (v) {
                            const int64_t xs = x + (__builtin_ctzll(v) >> 1);          // first base of the event inside a run of the reference
                            if (xs >= k - 1 && xs + k <= (int64_t)pc.ref_len) keep = false; else walk = true;
                            break;
                        }
                    }
                }
                if (walk) {
                    const int64_t lo = ty == NS_INS ? pos - 1 : pos, hi = pos + len - 1;
                    // [lo, hi] overlaps a run of >= k bases iff the run holds lo, hi or — lying strictly inside — one of lo + k, lo + 2k, ...
                    for (int64_t x = lo; keep; x = min(x + k, hi)) {
                        keep = win8 ? !in_hp_run_win8(A.ref, pc, key, a, x, k) : win ? !in_hp_run_win(A.ref, pc, key, a, x, k) : !in_hp_run(A.ref, pc, key, a, x, k);
                        if (x >= hi) break;
                    }
                }
                const uint64_t km = __ballot(keep);
                const uint32_t before = (uint32_t)__popcll(km & ((1ull << lane) - 1ull));
                int32_t d = keep ? (ty == NS_INS ? (int32_t)len : ty == NS_DEL ? -(int32_t)len : 0) : 0;
                const int32_t incl = (int32_t)wave_incl_scan((uint32_t)d);   // inclusive prefix sum of the length changes
                if (keep) {
                    ns_event o; o.pos = e.pos; o.info = ns_ev_pack((uint32_t)len, ty, shift + incl - d);
                    ev[w + before] = o;                                    // w + before <= j: never ahead of the batch being read
                    err_len += nl + dec_digits(e.pos) + dec_digits((uint32_t)len) + 2u * (uint32_t)len + 9u;
                }
                w += (uint32_t)__popcll(km);
                shift += __builtin_amdgcn_readlane(incl, 63);
            }
            p.n_ev = w; p.out_len = (uint32_t)((int32_t)p.ref_len + shift);
            if (lane == 0) A.pieces[rd.piece_off + pi] = p;
        }
        seq_len += p.out_len; scr_len += p.out_len;
    }
    err_len = wave_sum(err_len);
    if (A.polya) seq_len += A.polya[r];             // transcriptome: the polyA tail sits between the segment and the tail
    if (lane == 0) {
        rd.seq_len = (uint32_t)seq_len;             // length before mutate_homo
        A.reads[r] = rd;
        A.scr_len[r] = scr_len;                     // the scratch buffer holds the pieces
        A.err_len[r] = A.prm.emit_errlog ? err_len : 0;
    }
}

// final length check after the homopolymer stage (S:1429-1430 / metagenome S:1023-1024) and the record size of an accepted read
__device__ inline void hp_final_length(const GenArgs &A, uint64_t r, ns_read &rd, uint32_t a, uint64_t final_len, unsigned long long &st_bases,
                                       unsigned long long &st_fail) {
    const bool trx_al = A.prm.trx && A.prm.kind != NS_KIND_UNALIGNED;       // no length limits on aligned transcriptome reads
    if (A.meta && !A.key_pos && A.prm.kind != NS_KIND_UNALIGNED) {           // a pass of a metagenome worker (S:1023-1024): the
        if ((int64_t)final_len < A.prm.min_len || (int64_t)final_len > A.prm.max_len) {   // read is not accepted by this pass
            A.accept[r] = 0; st_fail = 1;
            unsigned long long rb = 0;               // k_chain had counted it as accepted
            for (uint32_t pi = 0; pi < rd.n_pieces; ++pi) rb += A.pieces[rd.piece_off + pi].ref_len;
            atomicAdd(&A.stats[2], 0ull - rb);
            atomicAdd(&A.stats[3], 0ull - (unsigned long long)A.sort_key[r]);
        } else {
            rd.seq_len = (uint32_t)final_len;
            A.reads[r] = rd;
            A.rec_len[r] = A.prm.emit_records ? (uint64_t)A.name_len[r] + 2 + final_len + 1 + (A.prm.fastq ? final_len + 3 : 0) : 0;
        }
    } else
    if (!trx_al && ((int64_t)final_len < A.prm.min_len || (int64_t)final_len > A.prm.max_len)) {      // S:1429-1430
        const uint32_t epoch = (A.rstate[r] & 0xffffu) + 1;
        A.rstate[r] = epoch & 0xffffu;
        A.att_base[r] = a + 1;
        st_fail = 1;
    } else {
        rd.seq_len = (uint32_t)final_len;
        A.reads[r] = rd;
        A.rec_len[r] = A.prm.emit_records ? (uint64_t)A.name_len[r] + 2 + final_len + 1 + (A.prm.fastq ? final_len + 3 : 0) : 0;
        st_bases = final_len;
    }
}

// mutate_homo (S:618-705) as an EDIT LIST over the pre-homopolymer read in the scratch buffer, one read per wavefront, in two kernels
// (as one kernel the drain's registers — fp64 norminv, Philox — held the scan at 4 wavefronts per SIMD):
//
// k_hp_scan — the run scan (S:627-637) streams a piece 1024 bases at a time, 16 per lane, and looks BACKWARDS only: a base that differs
// from its predecessor starts a run and thereby closes the previous one, whose start is the nearest run start before it — in the lane's
// own chunk, in the nearest lower lane that has one (ballot + one cross-lane read), or carried over from the earlier tiles in an SGPR.
// So there is no look-ahead, no dependent load, and the loads of the next tiles are in flight while a tile is scanned.  The runs of
// >= k bases are staged in an LDS list and leave for the piece's slots of the run buffer in coalesced groups.
//
// k_hp_drain — one lane per run: new length (S:644-665), the mismatches among the new bases (S:668-684), and from them the events the
// second record pass applies (materialise_piece, MAT_HP_FINAL): a deletion for a contraction, insertions of <= 15 letters for an
// expansion, one-base substitutions.  Wavefront prefix sums give every run its event slots and its cumulative shift.  Also: the final
// length of every piece and of the read (checked by k_hp_finalize, S:1429-1430).
#define NS_HPC_LIST 544u          // 4.9 KB per wavefront: eight wavefronts per SIMD
struct HpCountLds { uint32_t s0[NS_HPC_LIST], len[NS_HPC_LIST]; uint8_t base[NS_HPC_LIST]; };
__device__ __forceinline__ uint4 hp_load16(const uint8_t *__restrict__ sq, uint32_t n, uint32_t c) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (c < n) __builtin_memcpy(&v, sq + c, 16);                      // (the scratch buffer has slack behind the last read)
    return v;
}
// a run in the run buffer: start in the piece, length << 8 | base
__global__ void __launch_bounds__(64 * NS_WPB) k_hp_scan(GenArgs A, uint2 *__restrict__ runs, uint32_t *__restrict__ n_runs) {
    __shared__ HpCountLds list_lds[NS_WPB];
    HpCountLds &R = list_lds[threadIdx.x >> 6];
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t r = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= A.prm.n_reads) return;
    const ns_read rd = A.reads[r];
    if (rd.flags) return;
    const uint32_t k = A.prm.kmer_bias;
    const uint64_t scr_off = uni64(A.scr_off[r]);
    bool over = false;
    uint64_t q = 0;
    for (uint32_t pi = 0; pi < rd.n_pieces; ++pi) {
        const uint32_t gp = rd.piece_off + pi;
        const ns_piece p = A.pieces[gp];
        const uint32_t n = uni(p.out_len);
        if (!uni(p.kind)) {
            const uint8_t *sq = A.scr + scr_off + q;
            const uint32_t ord = uni(A.hp_pord[r]) + pi;
            const uint64_t ev0 = hp_ev_slot(A, scr_off + q, ord);
            const uint64_t cap64 = hp_ev_slot(A, scr_off + q + n, ord + 1) - ev0;
            const uint32_t cap = cap64 > 0xffffffffull ? 0xffffffffu : (uint32_t)cap64;
            uint2 *dst = runs + ev0;
            uint32_t n_out = 0;                                                  // runs of the piece written so far (wave-uniform)
            uint32_t n_list = 0;                                                 // runs waiting in the list (wave-uniform)
            // long runs one tile can close: 16 / k + 1 per chunk for k >= 4, one per base below
            const uint32_t tile_max = k >= 4 ? 64u * (16u / k + 1u) : 1024u;
            const bool direct = tile_max + 1u > NS_HPC_LIST;                    // k < 4: a tile can close more runs than the list holds — they go straight to the run buffer
            uint32_t carry_byte = 0xffu;                                         // base in front of the tile (0xff: none)
            uint32_t open_start = 0;                                             // start of the run that is open where the tile begins
            uint32_t t0 = 0;
            do {
                uint4 v0 = hp_load16(sq, n, t0 + 16 * lane), v1 = hp_load16(sq, n, t0 + 1024 + 16 * lane), v2 = hp_load16(sq, n, t0 + 2048 + 16 * lane);
                for (; t0 < n && (direct || n_list + tile_max + 1u <= NS_HPC_LIST); t0 += 1024) {
                    const uint4 v = v0;
                    v0 = v1; v1 = v2; v2 = hp_load16(sq, n, t0 + 3072 + 16 * lane);   // three tiles ahead
                    const uint32_t c = t0 + 16 * lane;
                    const uint32_t valid = c >= n ? 0u : min(16u, n - c);
                    // base in front of the chunk: the neighbouring lane's last base, for lane 0 the carry
                    const uint32_t pb = dpp_wave_shr1(carry_byte, v.w >> 24) & 0xffu;
                    const uint32_t p0 = v.x << 8 | pb, p1 = v.y << 8 | v.x >> 24, p2 = v.z << 8 | v.y >> 24, p3 = v.w << 8 | v.z >> 24;
                    // bit b: a run starts at base b of the chunk (the class bits of the bases do not take part in the comparison)
                    uint32_t M = movemask4(nonzero_bytes((v.x ^ p0) & NS_CLS_STRIP)) | movemask4(nonzero_bytes((v.y ^ p1) & NS_CLS_STRIP)) << 4 |
                                 movemask4(nonzero_bytes((v.z ^ p2) & NS_CLS_STRIP)) << 8 | movemask4(nonzero_bytes((v.w ^ p3) & NS_CLS_STRIP)) << 12;
                    M &= (1u << valid) - 1u;
                    // the nearest run start before the chunk: last start of the nearest lower lane that has one, else the carry
                    const uint64_t B = __ballot(M != 0);
                    const uint64_t below_l = B & ((1ull << lane) - 1ull);
                    const uint32_t last_own = c + 31u - (uint32_t)__clz((int)M);      // (garbage when M == 0: never read then)
                    const uint32_t from = below_l ? 63u - (uint32_t)__clzll((long long)below_l) : lane;
                    const uint32_t ps_l = (uint32_t)__shfl((int)last_own, (int)from);
                    const uint32_t ps = below_l ? ps_l : open_start;
                    const uint32_t back = c - ps;                                   // >= 1 for c > 0 (position 0 always starts a run)
                    // window: bit 16 + b = base b of the chunk, bits < 16 = the 16 bases in front of it
                    uint32_t W = M << 16;
                    if (c && back <= 16u) W |= 1u << (16u - back);
                    // a run ENDS in front of every start except position 0; it has >= k bases iff no start lies among the k - 1 positions
                    // before its end (a start further back than the window leaves the test true: the run is then longer than 16)
                    uint32_t E = M << 16;
                    if (c == 0) E &= ~(1u << 16);
                    if (k <= 16) {                                                   // E &= ~(W << 1 | ... | W << (k - 1)): the starts smeared upwards by doubling
                        uint32_t sm = W, have = 1;                                   // (sm = W | W << 1 | ... | W << (have - 1); k - 1 shifts in ~log2 k steps)
                        while (2u * have <= k - 1u) { sm |= sm << have; have *= 2u; }
                        if (have < k - 1u) sm |= sm << (k - 1u - have);
                        if (k > 1u) E &= ~(sm << 1);
                    }
                    else E = (M && c && back + (uint32_t)__builtin_ctz(M) >= k) ? (M & (0u - M)) << 16 : 0u;   // only the run closed by the first start
                    // ---- the long runs that end in front of the window bits E go to the list
                    const uint32_t rc = (uint32_t)__builtin_popcount(E), incl = wave_incl_scan(rc);
                    uint32_t slot = (direct ? n_out : n_list) + incl - rc;
                    for (uint32_t Em = E; Em; Em &= Em - 1, ++slot) {
                        const uint32_t b = (uint32_t)__builtin_ctz(Em);                 // the run ends in front of window bit b
                        const uint32_t below = W & ((1u << b) - 1u);                      // its start: the nearest start before it —
                        const uint32_t st = below ? c - 16u + 31u - (uint32_t)__clz((int)below) : ps;   // in the window, or further back
                        // base of the run = the base in front of bit b: chunk byte b - 17, or the byte in front of the chunk
                        const uint32_t i = b - 17u;
                        const uint32_t wv = b == 16u ? pb : (i < 8 ? (i < 4 ? v.x : v.y) : (i < 12 ? v.z : v.w)) >> (8 * (i & 3));
                        if (direct) { if (slot < cap) dst[slot] = make_uint2(st, (c - 16u + b - st) << 8 | (wv & 0xd7u)); }
                        else { R.s0[slot] = st; R.len[slot] = c - 16u + b - st; R.base[slot] = (uint8_t)(wv & 0xd7u); }
                    }
                    if (direct) n_out += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                    else n_list += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                    // carries
                    if (B) open_start = (uint32_t)__shfl((int)last_own, 63 - __clzll((long long)B));
                    carry_byte = (uint32_t)__builtin_amdgcn_readlane((int)(v.w >> 24), 63);
                }
                if (t0 >= n && n && n - open_start >= k) {                       // the run that is open at the end of the piece
                    if (direct) {
                        if (lane == 0 && n_out < cap) dst[n_out] = make_uint2(open_start, (n - open_start) << 8 | (uint32_t)(sq[n - 1] & 0xd7u));
                        ++n_out;
                    } else {
                        if (lane == 0) { R.s0[n_list] = open_start; R.len[n_list] = n - open_start; R.base[n_list] = (uint8_t)(sq[n - 1] & 0xd7u); }
                        ++n_list;
                    }
                }
                // ---- the list leaves for the run buffer
                wave_sync();
                for (uint32_t j = lane; j < n_list; j += 64)
                    if (n_out + j < cap) dst[n_out + j] = make_uint2(R.s0[j], R.len[j] << 8 | R.base[j]);
                n_out += n_list;
                wave_sync();
                n_list = 0;
            } while (t0 < n);
            if (n_out > cap) over = true;
            if (lane == 0) n_runs[gp] = min(n_out, cap);
        }
        q += n;
    }
    if (lane == 0 && over) atomicAdd(&A.stats[7], 1ull);                        // a piece outgrew its capacity: the stage is repeated with more
}

#ifndef NS_HPD_WAVES
#define NS_HPD_WAVES 4
#endif
__global__ void __launch_bounds__(64 * NS_WPB, NS_HPD_WAVES) k_hp_drain(GenArgs A, const uint2 *__restrict__ runs, const uint32_t *__restrict__ n_runs,
                                                                       uint64_t *__restrict__ hp_final) {
    struct DrainLds { uint32_t pos[2][64], tl[2][64], wd[2][64]; };
    __shared__ DrainLds drain_lds[NS_WPB];
    DrainLds &S = drain_lds[threadIdx.x >> 6];
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t r = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= A.prm.n_reads) return;
    const ns_read rd = A.reads[r];
    if (rd.flags) return;
    const ns_key key = read_key(A, r);
    const uint32_t a = rd.attempts;
    const uint64_t scr_off = uni64(A.scr_off[r]);
    bool over = false;
    uint64_t q = 0, final_len = (uint64_t)rd.head + rd.tail + (A.polya ? A.polya[r] : 0u);
    for (uint32_t pi = 0; pi < rd.n_pieces; ++pi) {
        const uint32_t gp = rd.piece_off + pi;
        const ns_piece p = A.pieces[gp];
        const uint32_t n = uni(p.out_len);
        uint32_t flen = n;
        if (!uni(p.kind)) {
            const uint32_t sid = pi >> 1;
            const uint32_t ord = uni(A.hp_pord[r]) + pi;
            const uint64_t ev0 = hp_ev_slot(A, scr_off + q, ord);
            const uint64_t cap64 = hp_ev_slot(A, scr_off + q + n, ord + 1) - ev0;
            const uint32_t cap = cap64 > 0xffffffffull ? 0xffffffffu : (uint32_t)cap64;
            ns_event *ev = A.hp_ev + ev0;
            uint32_t *wd = A.hp_wd + ev0;
            const uint2 *src = runs + ev0;
            const uint32_t n_list = uni(n_runs[gp]);
            uint32_t n_ev = 0, shift = 0;                                      // events filed / cumulative length change (wave-uniform)
            for (uint32_t j0 = 0; j0 < n_list; j0 += 64) {
                const uint32_t j = j0 + lane;
                const bool on = j < n_list;
                uint32_t s0 = 0, L = 0, base = 'A', size = 0, ne = 0, slot = 0, sh = 0, d_incl = 0, ne_incl = 0;
                if (on) { const uint2 rn = src[j]; s0 = rn.x; L = rn.y >> 8; base = rn.y & 0xffu; size = hp_new_size(A.m, key, sid, a, s0, L, base); }
                // the run's edits: evaluated once — the first two (all that all but a few runs in a thousand have) wait in the lane's LDS
                // slots while wavefront prefix sums give the event slots and shifts — then filed; a run with more is evaluated again
                if (on) {
                    uint32_t seen = 0;
                    ne = hp_run_events(A.m.hp_mis_rate, key, sid, a, s0, L, size, base, [&](uint32_t pos, uint32_t ty, uint32_t len, uint32_t word) {
                        if (seen < 2u) { S.pos[seen][lane] = pos; S.tl[seen][lane] = ty << 12 | len; S.wd[seen][lane] = word; }
                        ++seen;
                    });
                }
                {
                    const uint32_t d = on ? size - L : 0u;                   // (mod 2^32)
                    ne_incl = wave_incl_scan(ne); d_incl = wave_incl_scan(d);
                    slot = n_ev + ne_incl - ne; sh = shift + d_incl - d;
                }
                if (on && ne && slot + ne <= cap) {
                    if (ne <= 2u) {
                        const uint32_t tl0 = S.tl[0][lane];
                        ns_event e; e.pos = S.pos[0][lane]; e.info = ns_ev_pack(tl0 & 0xfffu, tl0 >> 12, (int32_t)sh);
                        ev[slot] = e; wd[slot] = S.wd[0][lane];
                        if (ne == 2u) {
                            const uint32_t ty0 = tl0 >> 12, l0 = tl0 & 0xfffu, tl1 = S.tl[1][lane];
                            const uint32_t sh1 = ty0 == NS_INS ? sh + l0 : ty0 == NS_DEL ? sh - l0 : sh;
                            e.pos = S.pos[1][lane]; e.info = ns_ev_pack(tl1 & 0xfffu, tl1 >> 12, (int32_t)sh1);
                            ev[slot + 1] = e; wd[slot + 1] = S.wd[1][lane];
                        }
                    } else
                        hp_run_events(A.m.hp_mis_rate, key, sid, a, s0, L, size, base, [&](uint32_t pos, uint32_t ty, uint32_t len, uint32_t word) {
                            ns_event e; e.pos = pos; e.info = ns_ev_pack(len, ty, (int32_t)sh);
                            ev[slot] = e; wd[slot] = word; ++slot;
                            if (ty == NS_INS) sh += len; else if (ty == NS_DEL) sh -= len;
                        });
                }
                n_ev += (uint32_t)__builtin_amdgcn_readlane((int)ne_incl, 63);
                shift += (uint32_t)__builtin_amdgcn_readlane((int)d_incl, 63);
            }
            if (n_ev > cap) over = true;
            flen = n + shift;
            if (lane == 0) A.hp_nev[gp] = min(n_ev, cap);
        }
        if (lane == 0) A.hp_len[gp] = flen;
        final_len += flen;
        q += n;
    }
    if (lane == 0) {
        hp_final[r] = final_len;                                                // (checked by k_hp_finalize, S:1429-1430)
        if (over) atomicAdd(&A.stats[7], 1ull);                                 // a piece outgrew its event capacity: the stage is repeated with more
    }
}

// the final length check of every read (S:1429-1430 / metagenome S:1023-1024) once k_hp_scan / k_hp_drain have run without a capacity overflow
__global__ void __launch_bounds__(256) k_hp_finalize(GenArgs A, const uint64_t *__restrict__ final_len) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long st_bases = 0, st_fail = 0;
    if (r < A.prm.n_reads && !A.stats[7]) {
        ns_read rd = A.reads[r];
        if (!rd.flags) hp_final_length(A, r, rd, rd.attempts, final_len[r], st_bases, st_fail);
        A.scr_len[r] = st_bases;                    // emitted bases of the read: summed by k_sum_u64 afterwards
    }
    st_fail = wave_sum(st_fail);
    if ((threadIdx.x & 63) == 0 && st_fail) atomicAdd(&A.stats[5], st_fail);
}
// batches without records: the pieces report their emitted length, like the path without -k
__global__ void __launch_bounds__(256) k_hp_report(GenArgs A, uint64_t n_pieces) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_pieces && !A.pieces[i].kind) A.pieces[i].out_len = A.hp_len[i];
}

// sum of v[0 .. n) added to *dst: grid-stride, one atomic per wavefront
__global__ void __launch_bounds__(256) k_sum_u64(const uint64_t *__restrict__ v, uint64_t n, unsigned long long *dst) {
    unsigned long long acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) acc += v[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(dst, acc);
}

// size of a read's error-profile rows (what k_errlog will write), one read per wavefront, lane per event.  (As a loop over the events
// inside the thread-per-read chain kernel this cost 4.4 ms per 10^6 reads: every thread walked its own list.)
__global__ void __launch_bounds__(64 * NS_WPB) k_errlen(GenArgs A) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t r = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= A.prm.n_reads) return;
    const ns_read rd = A.reads[r];
    unsigned long long sum = 0;
    if (!rd.flags) {
        const uint32_t nl = A.name_len[r];
        for (uint32_t pi = 0; pi < rd.n_pieces; ++pi) {
            const ns_piece p = A.pieces[rd.piece_off + pi];
            if (p.kind) continue;                                    // gaps and unaligned reads have no rows (S:1556, 1501)
            const ns_event *ev = A.events + p.ev_off;
            for (uint32_t j = lane; j < p.n_ev; j += 64) {
                const ns_event e = ev[j];
                sum += nl + dec_digits(e.pos) + dec_digits(ns_ev_len(e.info)) + 2u * ns_ev_len(e.info) + 9u;
            }
        }
    }
    sum = wave_sum(sum);
    if (lane == 0) A.err_len[r] = sum;
}

// ---------------------------------------------------------------------------------------------------------
// k_errlog: error-profile rows "name\tpos\ttype\tlen\tref\tnew\n" in descending position order (S:1960, 2006-2008)
// ---------------------------------------------------------------------------------------------------------
// Round 4: the 64 rows of an iteration are assembled side by side in ONE LDS buffer and leave as aligned 16-byte stores, lane l the
// l-th chunk of the block — until then every lane stored its own ~67-byte row with five unaligned 16-byte stores, each instruction
// touching 64 different cache lines: the kernel was bound by the memory pipeline's address handling (11.5 ms per 950 000 reads for
// 17.7 KB of text per read = 0.18 of HBM), not by HBM.  A block of rows that does not fit the buffer (names beyond ~60 characters with
// long payloads) takes the row-per-lane stores.
// Round 5: every DS access is ALIGNED.  The counters showed the LDS pipe 76 % busy at 19 LDS cycles per DS instruction (5.5 in the record
// kernel), and scripts/microbench/lds_align.hip says why: a ds_write_b64 / ds_read_b64 / b32 / b16 whose address is not a multiple of
// its size costs ~58 cycles per CU against ~7 (profiles/r05/microbench_lds_align.log) — and a row starts at any byte.  So
//   * the block lies in LDS at the destination's offset inside its 16-byte chunk (buf + (dst & 15)): the copy-out is one aligned
//     ds_read_b128 and one aligned 16-byte global store per lane (it was two ds_read_b64 at any offset);
//   * the read name is kept FOUR times, copy s shifted by s bytes: a lane whose row starts s bytes behind a dword boundary copies
//     aligned dwords from copy s (it was six ds_write_b64 at any offset per row); the bytes of a dword a row shares with its neighbours
//     and the row's fields are byte stores (6.8 cycles each whatever the address).
// Round 5, second step: the kernel is bound by how many wavefronts a SIMD holds (same-box A/B: an 8 KB block buffer = 4 wavefronts per
// SIMD 8.63 ms, 4.5 KB = 6 wavefronts 7.46 ms), so the block buffer is a template parameter: BUF = 5 120 bytes when 64 average rows of the
// batch fit with four bytes to spare each (the host knows the batch's bytes per row; a block that still does not fit takes the
// row-per-lane stores), else 8 192.  What did not pay on top of it (profiles/r05/ab_errlog.log): staging a block that does not fit in lane
// groups, and fetching the event of the NEXT iteration and the reference bases under this one before the LDS work — 83-84 VGPRs = five
// wavefronts per SIMD: 7.63 ms; bounded to 80 VGPRs with 20 bytes of scratch: 7.7-7.9.
#define NS_ERR_BUF_SMALL 5120u
#define NS_ERR_BUF_LARGE 8192u
#define NS_ERR_PAD 16u            // the block starts at the destination's offset inside its 16-byte chunk
#define NS_ERR_NAME 256u          // the read name, once per read (four shifted copies)
#define NS_ERR_NAME_ROW (NS_ERR_NAME + 8u)
template <uint32_t BUF>
__global__ void __launch_bounds__(64) k_errlog(GenArgs A) {
    __shared__ __align__(16) uint8_t buf_lds[NS_ERR_PAD + BUF + 16];
    __shared__ __align__(16) uint8_t name_lds[4][NS_ERR_NAME_ROW];
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t r = blockIdx.x;
    if (r >= A.prm.n_reads) return;
    const ns_read rd = A.reads[r];
    if (rd.flags) return;
    const ns_key key = read_key(A, r);
    const uint32_t a = rd.attempts;
    const uint32_t nl = A.name_len[r];
    const uint8_t *name = A.records + rd.rec_off + 1;
    const bool name_in_lds = nl <= NS_ERR_NAME;
    if (name_in_lds) {                                              // (k_names is done: ns_generate orders the kernels)
        // copy s, dword j = name bytes 4 j - s .. 4 j - s + 3 (zero outside the name)
        for (uint32_t j = lane; 4u * j < nl + 4u; j += 64) {
            uint32_t v0 = 0;
            for (uint32_t b = 0; b < 4; ++b) if (4u * j + b < nl) v0 |= (uint32_t)name[4u * j + b] << (8 * b);
            const uint32_t vp = j ? (uint32_t)name[4u * j - 1] | (uint32_t)name[4u * j - 2] << 8 | (uint32_t)name[4u * j - 3] << 16 : 0u;   // bytes -1, -2, -3
            *reinterpret_cast<uint32_t *>(&name_lds[0][4u * j]) = v0;
            *reinterpret_cast<uint32_t *>(&name_lds[1][4u * j]) = v0 << 8 | (vp & 0xffu);
            *reinterpret_cast<uint32_t *>(&name_lds[2][4u * j]) = v0 << 16 | (vp & 0xffu) << 8 | ((vp >> 8) & 0xffu);
            *reinterpret_cast<uint32_t *>(&name_lds[3][4u * j]) = v0 << 24 | (vp & 0xffu) << 16 | ((vp >> 8) & 0xffu) << 8 | (vp >> 16);
        }
        wave_sync();
    }
    uint64_t base = A.err_off[r];
    for (uint32_t pi = 0; pi < rd.n_pieces; pi += 2) {
        const ns_piece p = A.pieces[rd.piece_off + pi];
        PieceCtx pc = load_piece(A.events, A.ref, p, pi);
        ns_event e_nx; e_nx.pos = 0; e_nx.info = 0;             // the events of the iteration to come: rows are written from the LAST event to the first
        if (lane < p.n_ev) e_nx = pc.ev[p.n_ev - 1 - lane];
        for (uint32_t j0 = 0; j0 < p.n_ev; j0 += 64) {
            // rows are written from the LAST event to the first
            const uint32_t k = j0 + lane;
            const bool active = k < p.n_ev;
            const ns_event e = e_nx;                                                        // (loaded in front of the last copy-out)
            const uint32_t len = ns_ev_len(e.info), ty = ns_ev_type(e.info);
            const uint32_t tail = dec_digits(e.pos) + dec_digits(len) + 2u * len + 9u;      // "\t<pos>\t<type>\t<len>\t<ref>\t<new>\n"
            const uint32_t row = active ? nl + tail : 0;
            const uint32_t incl = wave_incl_scan(row);                                      // row offsets: prefix sum over the wavefront
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            const bool staged = name_in_lds && total <= BUF;                               // wave-uniform
            // (P: LdsBytes for the staged block — volatile, so that the compiler cannot merge neighbouring byte stores into a wide store at
            // an odd address, and in the LDS address space by type, because volatile accesses are not inferred into it)
            auto fields = [&](auto w) __attribute__((always_inline)) {                   // the rest of the row at w
                if (A.dbg & (1u << 17)) return;                           // (NS_DEBUG_SKIP, profiling only — 1 << 16: no name copy, 17: no fields, 18: no letter
                                                                          // columns, 19: no copy-out; profiles/r05/ablate_errlog.log)
                *w++ = '\t'; w = put_dec_p(w, e.pos); *w++ = '\t';
                const char *tn = ty == NS_MIS ? "mis" : ty == NS_INS ? "ins" : "del";
                *w++ = (uint8_t)tn[0]; *w++ = (uint8_t)tn[1]; *w++ = (uint8_t)tn[2];
                *w++ = '\t'; w = put_dec_p(w, len); *w++ = '\t';
                auto w2 = w + len + 1;
                if (A.dbg & (1u << 18)) return;
                // The two letter columns.  Round 5: as straight-line code for what all but one event in thousands is — <= 16 letters, plain
                // bases under them, not across the origin: the 16 reference bytes in ONE unaligned load, the letters of the event from its
                // ONE word (payload_word: 2-bit fields / successive base-3 digits) four at a time through byte permutes (the record kernel's
                // forms: S:1990 / S:1968-1972), then one pass of byte stores that ends with the longest run of the WAVEFRONT.  The
                // per-byte loop this replaces ran as often, but with ~200 instructions of divergent control flow per trip (the profile's
                // 2 094 scalar instructions per read were its exec-mask bookkeeping); it stays for the events the fast form does not take.
                const bool need_ref = ty != NS_INS;
                const bool inside = pc.pos + e.pos + 16ull <= pc.chrom_len;
                uint32_t r16[4] = {0x2d2d2d2du, 0x2d2d2d2du, 0x2d2d2d2du, 0x2d2d2d2du}, n16[4] = {0x2d2d2d2du, 0x2d2d2d2du, 0x2d2d2d2du, 0x2d2d2d2du};   // '-'
                if (need_ref && inside) __builtin_memcpy(r16, A.ref.bases + pc.chrom_base + pc.pos + e.pos, 16);   // (the engine's copy of the reference is padded)
                bool fast = len <= 16u && (!need_ref || inside);
                if (fast && need_ref) {                                   // an IUPAC code under the event: case_convert draws its base (resolve_base)
                    const uint32_t nb = 8u * (len & 3u), full = len >> 2;
                    uint32_t amb = 0;
#pragma unroll
                    for (uint32_t g = 0; g < 4; ++g) amb |= r16[g] & (g < full ? 0x80808080u : g == full ? (0x80808080u & ((1u << nb) - 1u)) : 0u);
                    fast = amb == 0;
                }
                if (ty != NS_DEL) {
                    const uint32_t word = payload_word(key, pc.sid, a, p.n_ev - 1 - k, 0);
                    uint32_t f3 = word;
#pragma unroll
                    for (uint32_t g = 0; g < 4; ++g) {
                        if (ty == NS_INS) {
                            const uint32_t x8 = (word >> (8u * g)) & 0xffu, t8 = (x8 | x8 << 12) & 0x000f000fu;
                            n16[g] = __builtin_amdgcn_perm(0u, 0x47435441u, (t8 | t8 << 6) & 0x03030303u);                        // S:1990
                        } else {
                            const uint32_t d0 = next_digit3(f3), d1 = next_digit3(f3), d2 = next_digit3(f3), d3 = next_digit3(f3);
                            const uint32_t d4 = d0 | d1 << 8 | d2 << 16 | d3 << 24;
                            const uint32_t vv = (r16[g] >> 1) & 0x03030303u;                              // A 0, C 1, T 2, G 3
                            const uint32_t rank4 = (vv & 0x01010101u) << 1 | ((vv >> 1) & 0x01010101u);  // rank in "ATCG": A 0, T 1, C 2, G 3
                            const uint32_t ge = ((d4 | 0x80808080u) - rank4) & 0x80808080u;              // per byte: digit >= rank
                            n16[g] = __builtin_amdgcn_perm(0u, 0x47435441u, d4 + (ge >> 7));                                       // S:1968-1972
                        }
                    }
                }
#pragma unroll
                for (uint32_t i = 0; i < 16; ++i) {
                    const bool on = fast && i < len;
                    if (!__ballot(on)) break;                             // (wave-uniform: no lane of the wavefront has a letter i; a ballot counts active lanes only)
                    if (on) { w[i] = (uint8_t)(r16[i >> 2] >> (8u * (i & 3u))); w2[i] = (uint8_t)(n16[i >> 2] >> (8u * (i & 3u))); }
                }
                if (!fast) {
                    uint32_t frac = 0;
                    for (uint32_t i = 0; i < len; ++i) {
                        if (ty != NS_DEL && !(i & 15u)) frac = payload_word(key, pc.sid, a, p.n_ev - 1 - k, i >> 4);
                        if (ty == NS_INS) { w[i] = '-'; w2[i] = bases_atcg((frac >> (2u * (i & 15u))) & 3u); }
                        else {
                            uint32_t x = e.pos + i;
                            uint8_t cur = resolve_base((uint32_t)ref_base_at(A.ref, pc, x), key, pc.sid, a, x);
                            w[i] = cur;
                            w2[i] = (ty == NS_MIS) ? mis_from_digit(cur, next_digit3(frac)) : (uint8_t)'-';
                        }
                    }
                }
                w[len] = '\t';
                w2[len] = '\n';
            };
            if (staged) {
                uint8_t *const dst0 = A.errlog + base;
                const uint32_t mis = (uint32_t)(uintptr_t)dst0 & 15u;                        // block byte i lies at buf_lds[mis + i]: chunk-congruent with dst0
                if (active) {
                    const uint32_t o = mis + (incl - row);                                  // the row starts at buf_lds[o]
                    const uint32_t al = o & 3u;                                             // ... al bytes behind a dword boundary (buf_lds is 16-byte aligned)
                    const LdsBytes q = (LdsBytes)buf_lds + o;
                    const LdsBytes src = (LdsBytes)name_lds[al];                            // copy al: name byte i at src[al + i]
                    const LdsWords qd = (LdsWords)((LdsBytes)buf_lds + (o - al));
                    const uint32_t d1 = (al + nl) >> 2;                                     // dwords [d0, d1) of the row's grid hold name bytes only
                    const uint32_t d0 = al ? 1u : 0u;
                    if (A.dbg & (1u << 16)) {}
                    else if (d1 > d0) {
                        const LdsWords sw = (LdsWords)src;
                        const uint32_t w_head = sw[0], w_tail = sw[d1];                     // the dwords the row shares with its neighbours: byte stores
                        for (uint32_t j = d0; j < d1; j += 4) {                             // four dwords in flight (volatile accesses keep their order)
                            const uint32_t a0 = sw[j], a1 = sw[min(j + 1u, d1 - 1u)], a2 = sw[min(j + 2u, d1 - 1u)], a3 = sw[min(j + 3u, d1 - 1u)];
                            qd[j] = a0;
                            if (j + 1u < d1) qd[j + 1u] = a1;
                            if (j + 2u < d1) qd[j + 2u] = a2;
                            if (j + 3u < d1) qd[j + 3u] = a3;
                        }
                        for (uint32_t b = al; b < 4u && al; ++b) q[b - al] = (uint8_t)(w_head >> (8u * b));
                        for (uint32_t b = 4u * d1; b < al + nl; ++b) q[b - al] = (uint8_t)(w_tail >> (8u * (b & 3u)));
                    } else for (uint32_t i = 0; i < nl; ++i) q[i] = src[al + i];
                    fields(q + nl);
                }
                wave_sync();
                // the block leaves: lane c the 16 bytes at errlog address (dst0 - mis) + 16 c = buf_lds[16 c .. 16 c + 15]
                uint8_t *const dstA = dst0 - mis;
                // the NEXT iteration's events are asked for in front of the stores, not behind them (the load then does not queue up behind
                // 4 KB of this wavefront's own writes: 6.34 -> 6.09 ms on top of the nontemporal stores, profiles/r05/ab_errlog_nt.log)
                e_nx.pos = 0; e_nx.info = 0; if (k + 64u < p.n_ev) e_nx = pc.ev[p.n_ev - 1 - (k + 64u)];
                for (uint32_t c = lane; 16u * c < mis + total && !(A.dbg & (1u << 19)); c += 64) {
                    const uint4 v = *reinterpret_cast<const uint4 *>(buf_lds + 16u * c);
                    const uint32_t s0 = c ? 0u : mis;                                           // bytes of the chunk in front of the block
                    const uint32_t e0 = min(16u, mis + total - 16u * c);                        // ... and where the block ends inside it
                    if (s0 == 0 && e0 == 16u) __builtin_nontemporal_store(ns_v4u_any{v.x, v.y, v.z, v.w}, reinterpret_cast<ns_v4u_any *>(dstA + 16u * c));   // (store16)
                    else {
                        uint64_t v0 = (uint64_t)v.x | (uint64_t)v.y << 32, v1 = (uint64_t)v.z | (uint64_t)v.w << 32;
                        shift_down_bytes(v0, v1, s0); store16(dstA + 16u * c + s0, e0 - s0, v0, v1);
                    }
                }
                wave_sync();
            } else if (active) {
                uint8_t *q = A.errlog + base + (incl - row);
                if (nl >= 16) {                                          // the read name, 16 bytes at a time (the last chunk overlaps)
                    for (uint32_t i = 0; i + 16 <= nl; i += 16) { uint4 v; __builtin_memcpy(&v, name + i, 16); __builtin_memcpy(q + i, &v, 16); }
                    if (nl & 15u) { uint4 v; __builtin_memcpy(&v, name + nl - 16, 16); __builtin_memcpy(q + nl - 16, &v, 16); }
                } else for (uint32_t i = 0; i < nl; ++i) q[i] = name[i];
                fields(q + nl);
            }
            if (!staged) { e_nx.pos = 0; e_nx.info = 0; if (k + 64u < p.n_ev) e_nx = pc.ev[p.n_ev - 1 - (k + 64u)]; }
            base += total;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// k_cs_hist: the counting loop of the characterisation stage (ns_cs_hist.h; src/besthit_to_histogram.py:316-365), one alignment per
// thread.  The 1-D histograms and the transition counters are privatised per workgroup in LDS (their hot bins — one-base mismatches,
// short matches — would serialise millions of atomics on a few addresses) and flushed once; the (previous match, next match) matrix is
// large and sparse: global atomics.
// ---------------------------------------------------------------------------------------------------------
struct CsHistDev {
    unsigned long long *dic;          // [5][1001]
    unsigned long long *err;          // [18] error_list, [3] first_error behind it
    unsigned long long *misc;         // [0] max_match [1] match_list overflow [2] `=` items
    unsigned long long *m2; uint32_t cap2;
};
#define NS_CSH_LDS_WORDS (5u * 1001u + 24u + 1u)
struct CsAccDev {
    uint32_t *l;                      // the workgroup's LDS counters: dic[5][1001], err[18], first[3], (3 spare), the largest match
    const CsHistDev *H;
    uint32_t mx;                      // largest length this thread handed to add_match
    __device__ __forceinline__ void d1(uint32_t which, uint32_t v) { if (v <= NS_CS_DICT_MAX) atomicAdd(&l[which * 1001u + v], 1u); }
    __device__ __forceinline__ void m2(uint32_t p, uint32_t s) {
        const uint32_t m = p > s ? p : s;
        mx = mx > m ? mx : m;
        if (H->m2 && m < H->cap2) atomicAdd(&H->m2[(uint64_t)p * H->cap2 + s], 1ull);
        else atomicAdd(&H->misc[1], 1ull);
    }
    __device__ __forceinline__ void err(uint32_t i) { atomicAdd(&l[5u * 1001u + i], 1u); }
    __device__ __forceinline__ void first(uint32_t i) { atomicAdd(&l[5u * 1001u + 18u + i], 1u); }
    __device__ __forceinline__ void skip() { atomicAdd(&H->misc[2], 1ull); }
};
__global__ void __launch_bounds__(256) k_cs_len(const uint64_t *__restrict__ off, uint32_t n_aln, uint32_t *__restrict__ key, uint32_t *__restrict__ idx) {
    const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n_aln) return;
    const uint64_t n = off[a + 1] - off[a];
    key[a] = n > 0xffffffffull ? 0xffffffffu : (uint32_t)n; idx[a] = a;
}
// qry != nullptr: the MAF branch — cs holds the reference lines, qry the query lines (maf_hist_alignment)
__global__ void __launch_bounds__(256) k_cs_hist(const uint8_t *__restrict__ cs, const uint64_t *__restrict__ off, uint32_t n_aln, CsHistDev H,
                                                 const uint32_t *__restrict__ order, const uint8_t *__restrict__ qry) {
    __shared__ uint32_t cnt[NS_CSH_LDS_WORDS];
    for (uint32_t i = threadIdx.x; i < NS_CSH_LDS_WORDS; i += blockDim.x) cnt[i] = 0;
    __syncthreads();
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid < n_aln) {
        const uint64_t a = order ? order[tid] : tid;          // visited by descending length: the 64 walks of a wavefront have similar trip counts
        const uint8_t *s = cs + off[a];
        const uint64_t n = off[a + 1] - off[a];
        // prev_match is only read before this alignment assigns it when its first op is an error: then it is what the alignments in
        // front of it left (the reference never resets it between alignments)
        CsAccDev acc{cnt, &H, 0u};
        if (qry) maf_hist_alignment(s, qry + off[a], n, acc);
        else {
            uint32_t pm = 0;
            CsBytes sb(s);                                   // (an 8-byte register window over the thread's string: ns_cs_hist.h)
            { CsCursor c; cs_cursor_init(c); int t; uint32_t l; if (cs_next_op(sb, n, c, t, l) && t != CS_MATCH) pm = cs_carry_in(cs, off, a); }
            cs_hist_alignment(sb, n, pm, nullptr, acc);
        }
        if (acc.mx) atomicMax(&cnt[NS_CSH_LDS_WORDS - 1u], acc.mx);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < NS_CSH_LDS_WORDS; i += blockDim.x) {
        const uint32_t v = cnt[i];
        if (!v) continue;
        if (i < 5u * 1001u) atomicAdd(&H.dic[i], (unsigned long long)v);
        else if (i < NS_CSH_LDS_WORDS - 1u) atomicAdd(&H.err[i - 5u * 1001u], (unsigned long long)v);
        else atomicMax(&H.misc[0], (unsigned long long)v);          // one atomic per workgroup for the largest match
    }
}

// ---------------------------------------------------------------------------------------------------------
// reference normalisation (once per ns_set_reference): upper-case, non-IUPAC -> N
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_normalise(uint8_t *bases, uint64_t n) {
    uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
    if (i >= n) return;
    if (i + 16 <= n) {
        uint4 v = *reinterpret_cast<uint4 *>(bases + i);
        uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t o = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) o |= (uint32_t)normalise_base((w[k] >> (8 * b)) & 0xff) << (8 * b);
            w[k] = o;
        }
        *reinterpret_cast<uint4 *>(bases + i) = make_uint4(w[0], w[1], w[2], w[3]);
    } else {
        for (uint64_t k = i; k < n; ++k) bases[k] = normalise_base(bases[k]);
    }
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

struct ns_ctx {
    int device = 0;
    hipStream_t stream = nullptr, stream2 = nullptr;   // stream2: cooperative chain of the longest reads, concurrent with the bulk
    hipStream_t stream3 = nullptr;                     // chimeric batches: the reads of several pieces, a thread per piece, next to both (created at first use)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_join3 = nullptr;
    std::string err;
    bool has_model = false, has_ref = false, has_batch = false;
    DevModel m{};
    DevRef ref{};
    std::vector<void *> model_allocs;
    void *ref_bases_owned = nullptr;
    std::vector<void *> ref_allocs;
    double cap_rate = 0.1;
    uint64_t ref_nbases = 0;
    uint32_t dbg = 0;          // NS_DEBUG_SKIP: phase-ablation bits for profiling only (results are wrong when set)
    uint32_t coop_min = 16384, coop_shift = 10;  // cooperative chain for the longest n>>shift reads of batches >= min (env: NS_COOP_MIN, NS_COOP_SHIFT);
                                                 // 10^6 reads, chain ms at shift 9 / 10 / 11 / 12: 4.18 / 3.63 / 3.90 / 4.32
    bool ucoop_lds = true;                       // ... with the run-length tables in LDS (k_chain<true, true>; env NS_UCOOP_LDS=0: from global memory, as until round 5)
    uint32_t ucoop_shift = 0;                    // unaligned reads: the longest n>>shift of a batch take the wave-per-read list, the rest the thread-per-read one (env: NS_UCOOP_SHIFT; 0: all)
    // planning + result buffers
    DevBuf l_cap, l_off, p_need, p_off;
    DevBuf n_pieces, piece_off, ev_cap, ev_off, rec_len, rec_off, err_len, err_off, name_len;
    DevBuf reads, pieces, events, stats, scan_tmp;
    DevBuf rec_slot[2], err_slot[2];   // two result slots: the record / error-profile images of the last batch and of the one before (ns_io.h)
    int slot = 0;                      // slot of the last batch
    IoEngine *io = nullptr;            // copy stream, staging slices, writer threads (created by the first ns_sink_open)
    std::vector<ns_sink *> sinks;
    DevBuf ord_bins;                 // k_order_*: histogram + cursors of the visiting-order bins
    DevBuf sort_key, sort_idx, sort_key_out, order, list_b, list_c, rstate, att_base, scr, scr_len, scr_off, hp_len, hp_nev, hp_ev, hp_wd, hp_runs, hp_nrun, slow_q, cls, hp_bm, hp_pcnt, hp_pord;
    uint32_t hp_shift = 5, hp_pad = 64, hp_cap_k = 0;       // -k: event capacity of a piece (hp_ev_slot), planned for kmer_bias hp_cap_k
    // metagenome: species view of the reference, abundances of the sample, per-pass scratch
    DevBuf species_chrom_off, t_reads, t_pieces, t_name_len, t_rec_len, t_err_len, accept, accept_scan, key_pos, draw_x, m_segptr,
        m_len, m_species, species_bases;
    DevBuf draw_sel, draw_sorted, meta_words, meta_num;
    DevBuf trx_chrom, trx_cum, trx_polya, polya;            // transcriptome: expression view of the reference, polyA length per read
    DevBuf trx_pick_e, trx_pick_y, trx_keys, trx_keys2, trx_prev, trx_cand, t_polya, t_ir_need;   // ... the pick walk of its aligned batches (trx_passes)
    uint32_t trx_margin = 128, trx_pick_pct = 125;          // candidates per block beyond NS_TRX_BLOCK / picks per candidate in percent: grown on demand, kept
    DevTrx tx{};
    bool has_trx = false;
    DevIr ir{};                                             // intron retention: genome, transcript structures, Markov chain
    bool has_ir = false;
    std::vector<void *> ir_allocs;
    DevBuf ir_need, ir_off, spliced;
    uint64_t spliced_bytes = 0;
    uint8_t *pin_small = nullptr;    // page-locked slots for the scalar read-backs of a call (read_small)
    struct PinBuf { void *p = nullptr; size_t cap = 0; } pin_a, pin_b, pin_c, pin_d;     // pinned host staging of the metagenome passes
    uint32_t nspecies = 0;
    bool has_abun = false, has_inflated = false, has_key_pos = false;
    std::vector<double> abun, abun_inflated, last_species_bases;
    bool lds_tables = false, coop_ok = false;
    uint32_t chain_block = NS_CHAIN_BLOCK;     // threads per workgroup of k_chain<LDS> (ns_load_model: 256, or 640 for a large image)
    size_t lds_bytes = 0;
    uint32_t hp_bm_k = 0;                // -k: the k the bitmap hp_bm was built for (0: none)
    ns_batch_info last{};
    hipEvent_t evt[16]{};
    bool sq_zeroed = false;                      // k_stats_fold has zeroed the slow-tile queue's counter and nothing has used the queue since
    bool rec_timed = false;                      // evt[12] / evt[13] bracket the record kernel of this call
    bool evt_ok = false;
    // ns_generate_step: the companion context the unaligned worker call of a step runs on (it borrows this context's reference, model and
    // mode tables) and the worker thread that makes that call
    ns_ctx *companion = nullptr;
    bool borrowed = false;            // this context IS a companion: the tables it points at belong to its owner
    ns_ctx *owner = nullptr;          // ... which is this one: every call on the companion takes the owner's tables as they are NOW (lend_tables)
    // the step gate: the companion holds its first chain launch until the owner's aligned call has launched its own chain — the aligned
    // call's planning kernels then run in 0.29 ms instead of 0.72 ms behind the unaligned chain's grid (same box: 9.9-10.1 -> 9.6-9.85 ms
    // per step, profiles/r05/ab_step_gate.log; NS_STEP_GATE=0: off).  Creating the companion's streams with the device's highest priority
    // changed nothing (9.57 / 9.67 against 9.60 / 9.67).
    std::atomic<int> gate{0};
    std::atomic<int> *gate_signal = nullptr, *gate_wait = nullptr;
    struct StepWorker;
    StepWorker *step = nullptr;
};
struct ns_ctx::StepWorker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    const ns_params *prm = nullptr;   // the posted call (nullptr: none)
    ns_batch_info *info = nullptr;
    int rc = 0;
    bool done = false, quit = false;
};

static int fail(ns_ctx *c, int code, const std::string &msg);
// the tables of a step companion (ns_generate_step) are its owner's
#define NS_NOT_ON_COMPANION(c) do { if ((c) && (c)->borrowed) return fail((c), NS_ESTATE, "this is a step companion: set reference, model and mode tables on the context that owns it"); } while (0)
static int fail(ns_ctx *c, int code, const std::string &msg) {
    if (c) c->err = msg;
    return code;
}
#define HIPCHK(call)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) return fail(ctx, NS_EHIP, std::string(#call ": ") + hipGetErrorString(e_)); \
    } while (0)

static int ensure(ns_ctx *ctx, DevBuf &b, size_t bytes) {
    if (bytes <= b.cap) return NS_OK;
    size_t want = bytes + bytes / 8 + 4096;
    if (b.p) { hipError_t e = hipFree(b.p); (void)e; b.p = nullptr; b.cap = 0; }
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        b.p = nullptr; b.cap = 0;
        size_t fr = 0, tot = 0;
        hipError_t e2 = hipMemGetInfo(&fr, &tot); (void)e2;
        (void)hipGetLastError();             // the failed allocation must not poison the next call
        return fail(ctx, NS_ENOMEM, std::string("hipMalloc(") + std::to_string(want) + " bytes): " + hipGetErrorString(e) + " (" +
                                        std::to_string(fr >> 20) + " MiB free of " + std::to_string(tot >> 20) + ")");
    }
    b.cap = want;
    return NS_OK;
}

static int ensure_pin(ns_ctx *ctx, ns_ctx::PinBuf &b, size_t bytes) {
    if (bytes <= b.cap) return NS_OK;
    size_t want = bytes + bytes / 8 + 4096;
    if (b.p) { hipError_t e = hipHostFree(b.p); (void)e; b.p = nullptr; b.cap = 0; }
    hipError_t e = hipHostMalloc(&b.p, want, hipHostMallocDefault);
    if (e != hipSuccess) { b.p = nullptr; return fail(ctx, NS_ENOMEM, std::string("hipHostMalloc: ") + hipGetErrorString(e)); }
    b.cap = want;
    return NS_OK;
}

// grows a buffer whose first `keep` bytes must survive
static int ensure_keep(ns_ctx *ctx, DevBuf &b, size_t bytes, size_t keep) {
    if (bytes <= b.cap) return NS_OK;
    size_t want = bytes + bytes / 2 + 4096;
    void *np = nullptr;
    hipError_t e = hipMalloc(&np, want);
    if (e != hipSuccess) return fail(ctx, NS_ENOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    if (b.p && keep) {
        e = hipMemcpyAsync(np, b.p, keep, hipMemcpyDeviceToDevice, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { hipError_t e2 = hipFree(np); (void)e2; return fail(ctx, NS_EHIP, std::string("hipMemcpy: ") + hipGetErrorString(e)); }
    }
    if (b.p) { e = hipFree(b.p); (void)e; }
    b.p = np; b.cap = want;
    return NS_OK;
}

template <typename T>
static int upload(ns_ctx *ctx, std::vector<void *> &pool, const T *src, size_t n, const T **dst) {
    *dst = nullptr;
    if (!n) return NS_OK;
    if (!src) return fail(ctx, NS_EINVAL, "null table pointer");
    void *p = nullptr;
    HIPCHK(hipMalloc(&p, n * sizeof(T)));
    pool.push_back(p);
    HIPCHK(hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice));
    *dst = static_cast<const T *>(p);
    return NS_OK;
}

extern "C" {

uint32_t ns_abi_version(void) { return NS_ABI_VERSION; }

// (Until round 6 a background context sent all but the longest eighth of its unaligned reads to the thread-per-read chain: the wave-per-read
// one was 2.65 ms of serialized atomics then and slowed the other call's chain.  At 1.45 ms it ends the unaligned call of a step after 3.6
// instead of 7.7 ms and leaves the record kernel alone — 0.53 instead of 0.48 of the roofline inside a step, the step itself 2 % longer:
// profiles/r06/ab_step_companion.log, call 39.  NS_UCOOP_SHIFT=3 brings the split back.)
int ns_set_background(ns_ctx *ctx, int on) {
    if (!ctx) return NS_EINVAL;
    (void)on;
    ctx->ucoop_shift = 0u;
    if (const char *d = getenv("NS_UCOOP_SHIFT")) ctx->ucoop_shift = (uint32_t)atoi(d) & 31u;
    return NS_OK;
}

// prio: 0 = the default stream priority, 1 / -1 = the highest / lowest the device offers (NS_STEP_PRIO, for the step companion's streams:
// an A/B knob).  Streams of another priority come from another pool of hardware queues — the runtime shares GPU_MAX_HW_QUEUES = 4 among the
// streams of ONE priority, so with the companion's streams on top of the owner's the two chain kernels of its call share a queue and run one
// after the other.  Measured (profiles/r06/ab_step_companion.log): high priority ends the unaligned call a millisecond earlier and the
// aligned one as much later, low priority starves it, eight queues at the default priority change nothing: the default stays.
static int create_stream(hipStream_t *s, int prio) {
    int least = 0, greatest = 0;
    if (prio && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least &&
        hipStreamCreateWithPriority(s, hipStreamNonBlocking, prio > 0 ? greatest : least) == hipSuccess) return 0;
    return hipStreamCreateWithFlags(s, hipStreamNonBlocking) == hipSuccess ? 0 : -1;
}
static int create_ctx(int device, ns_ctx **out, int prio);
int ns_create(int device, ns_ctx **out) { return create_ctx(device, out, 0); }
static int create_ctx(int device, ns_ctx **out, int prio) {
    if (!out) return NS_EINVAL;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return NS_ENODEV;
    if (device < 0 || device >= n) return NS_ENODEV;
    ns_ctx *ctx = new ns_ctx();
    ctx->device = device;
    // (ns_destroy releases whatever exists of a half-built context: null handles are skipped)
    if (hipSetDevice(device) != hipSuccess || create_stream(&ctx->stream, prio)) {
        ctx->stream = nullptr;
        ns_destroy(ctx);
        return NS_EHIP;
    }
    for (auto &e : ctx->evt) e = nullptr;
    bool ok = true;
    for (auto &e : ctx->evt)
        if (ok && hipEventCreate(&e) != hipSuccess) { e = nullptr; ok = false; }
    ctx->evt_ok = true;              // (the events that exist are destroyed with the context)
    if (!ok || create_stream(&ctx->stream2, prio) ||
        hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) != hipSuccess) { ns_destroy(ctx); return NS_EHIP; }
    if (hipHostMalloc((void **)&ctx->pin_small, 1024, hipHostMallocDefault) != hipSuccess) { ctx->pin_small = nullptr; ns_destroy(ctx); return NS_ENOMEM; }
    ctx->evt_ok = true;
    if (const char *d = getenv("NS_DEBUG_SKIP")) ctx->dbg = (uint32_t)atoi(d);
    if (const char *d = getenv("NS_COOP_MIN")) ctx->coop_min = (uint32_t)atoi(d);
    if (const char *d = getenv("NS_COOP_SHIFT")) ctx->coop_shift = (uint32_t)atoi(d) & 31u;
    if (const char *d = getenv("NS_UCOOP_SHIFT")) ctx->ucoop_shift = (uint32_t)atoi(d) & 31u;
    if (const char *d = getenv("NS_UCOOP_LDS")) ctx->ucoop_lds = atoi(d) != 0;
    *out = ctx;
    return NS_OK;
}

// Scalar read-backs of a call (totals of the scans, the counters): device -> page-locked slot -> destination, with the stream
// synchronised in between.  (hipMemcpyAsync into pageable memory goes through a staging blit kernel; next to another context's
// kernels on the same GPU that costs hundreds of microseconds per read-back.)
// the visiting order of a batch: reads by descending key (planned work) — bins of the key, 3 % wide (k_order_*); NS_EXACT_ORDER=1: a full sort (A/B)
static int visiting_order(ns_ctx *ctx, const uint32_t *keys, const uint32_t *idx, size_t n, uint32_t *list) {
    hipStream_t st = ctx->stream;
    int rc;
    if (getenv("NS_EXACT_ORDER")) {
        size_t tmp = 0;
        HIPCHK(hipcub::DeviceRadixSort::SortPairsDescending(nullptr, tmp, keys, (uint32_t *)ctx->sort_key_out.p, idx, list, (int)n, 0, 32, st));
        if ((rc = ensure(ctx, ctx->scan_tmp, tmp))) return rc;
        HIPCHK(hipcub::DeviceRadixSort::SortPairsDescending(ctx->scan_tmp.p, tmp, keys, (uint32_t *)ctx->sort_key_out.p, idx, list, (int)n, 0, 32, st));
        return NS_OK;
    }
    if (!ctx->ord_bins.p) {
        if ((rc = ensure(ctx, ctx->ord_bins, (2 * NS_ORD_BINS + 16) * 4))) return rc;
        HIPCHK(hipMemsetAsync(ctx->ord_bins.p, 0, (2 * NS_ORD_BINS + 16) * 4, st));
    }
    uint32_t *hist = (uint32_t *)ctx->ord_bins.p, *cursor = hist + NS_ORD_BINS;
    const unsigned tiles = (unsigned)((n + 1023) / 1024);
    if (!tiles) return NS_OK;
    k_order_hist<<<dim3(std::min(tiles, 256u)), dim3(1024), 0, st>>>(keys, (uint32_t)n, hist);
    k_order_scan<<<dim3(1), dim3(1024), 0, st>>>(hist, cursor);
    k_order_deal<<<dim3(tiles), dim3(1024), 0, st>>>(keys, (uint32_t)n, cursor, list);
    HIPCHK(hipGetLastError());
    return NS_OK;
}
static void fold_stats(ns_ctx *ctx, hipStream_t st) {
    uint32_t *z = ctx->slow_q.cap >= 16 ? (uint32_t *)ctx->slow_q.p : nullptr;
    k_stats_fold<<<dim3(1), dim3(64), 0, st>>>((unsigned long long *)ctx->stats.p, z);
    ctx->sq_zeroed = z != nullptr;
}
static int read_small(ns_ctx *ctx, hipStream_t st, void *dst, const void *src, size_t n, void *dst2 = nullptr, const void *src2 = nullptr, size_t n2 = 0) {
    if (n > (dst2 ? 512u : 1024u) || n2 > 512) return fail(ctx, NS_EINVAL, "read_small: too large");
    HIPCHK(hipMemcpyAsync(ctx->pin_small, src, n, hipMemcpyDeviceToHost, st));
    if (dst2) HIPCHK(hipMemcpyAsync(ctx->pin_small + 512, src2, n2, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    memcpy(dst, ctx->pin_small, n);
    if (dst2) memcpy(dst2, ctx->pin_small + 512, n2);
    return NS_OK;
}

static void free_pool(std::vector<void *> &pool) {
    for (void *p : pool) { hipError_t e = hipFree(p); (void)e; }
    pool.clear();
}

void ns_destroy(ns_ctx *ctx) {
    if (!ctx) return;
    if (ctx->step) {                                 // the step worker first: it may be inside a call on the companion
        { std::lock_guard<std::mutex> lk(ctx->step->mu); ctx->step->quit = true; }
        ctx->step->cv.notify_all();
        if (ctx->step->th.joinable()) ctx->step->th.join();
        delete ctx->step; ctx->step = nullptr;
    }
    if (ctx->companion) { ns_destroy(ctx->companion); ctx->companion = nullptr; }
    if (ctx->borrowed) {                             // a companion frees its own batch buffers only
        ctx->model_allocs.clear(); ctx->ref_allocs.clear(); ctx->ir_allocs.clear(); ctx->ref_bases_owned = nullptr;
        ctx->species_chrom_off = DevBuf{}; ctx->trx_chrom = DevBuf{}; ctx->trx_cum = DevBuf{}; ctx->trx_polya = DevBuf{};
    }
    hipError_t e = hipSetDevice(ctx->device); (void)e;
    if (ctx->io) { ctx->io->wait_all(); ctx->io->shutdown(); delete ctx->io; ctx->io = nullptr; }
    for (ns_sink *s : ctx->sinks) delete s;
    ctx->sinks.clear();
    if (ctx->stream) { e = hipStreamSynchronize(ctx->stream); e = hipStreamDestroy(ctx->stream); }
    if (ctx->stream2) { e = hipStreamSynchronize(ctx->stream2); e = hipStreamDestroy(ctx->stream2); }
    if (ctx->stream3) { e = hipStreamSynchronize(ctx->stream3); e = hipStreamDestroy(ctx->stream3); }
    if (ctx->ev_join3) e = hipEventDestroy(ctx->ev_join3);
    if (ctx->ev_fork) e = hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) e = hipEventDestroy(ctx->ev_join);
    free_pool(ctx->model_allocs);
    free_pool(ctx->ref_allocs);
    free_pool(ctx->ir_allocs);
    if (ctx->ref_bases_owned) e = hipFree(ctx->ref_bases_owned);
    DevBuf *bufs[] = {&ctx->n_pieces, &ctx->piece_off, &ctx->ev_cap, &ctx->ev_off, &ctx->rec_len, &ctx->rec_off,
                      &ctx->err_len, &ctx->err_off, &ctx->name_len, &ctx->reads, &ctx->pieces, &ctx->events,
                      &ctx->rec_slot[0], &ctx->rec_slot[1], &ctx->err_slot[0], &ctx->err_slot[1], &ctx->stats, &ctx->scan_tmp, &ctx->sort_key, &ctx->sort_idx, &ctx->ord_bins,
                      &ctx->sort_key_out, &ctx->order, &ctx->list_b, &ctx->list_c, &ctx->rstate, &ctx->att_base, &ctx->scr, &ctx->hp_nev, &ctx->hp_ev, &ctx->hp_wd, &ctx->hp_runs, &ctx->hp_nrun,
                      &ctx->scr_len, &ctx->scr_off, &ctx->hp_len, &ctx->slow_q, &ctx->l_cap, &ctx->l_off, &ctx->species_chrom_off, &ctx->t_reads, &ctx->t_pieces,
                      &ctx->t_name_len, &ctx->t_rec_len, &ctx->t_err_len, &ctx->accept, &ctx->accept_scan, &ctx->key_pos,
                      &ctx->draw_x, &ctx->m_segptr, &ctx->m_len, &ctx->m_species, &ctx->species_bases, &ctx->draw_sel,
                      &ctx->draw_sorted, &ctx->meta_words, &ctx->meta_num, &ctx->trx_chrom, &ctx->trx_cum, &ctx->trx_polya, &ctx->polya,
                      &ctx->ir_need, &ctx->ir_off, &ctx->spliced, &ctx->p_need, &ctx->p_off, &ctx->cls, &ctx->hp_bm, &ctx->hp_pcnt, &ctx->hp_pord,
                      &ctx->trx_pick_e, &ctx->trx_pick_y, &ctx->trx_keys, &ctx->trx_keys2, &ctx->trx_prev, &ctx->trx_cand, &ctx->t_polya, &ctx->t_ir_need};
    for (auto *pb : {&ctx->pin_a, &ctx->pin_b, &ctx->pin_c, &ctx->pin_d})
        if (pb->p) e = hipHostFree(pb->p);
    if (ctx->pin_small) e = hipHostFree(ctx->pin_small);
    for (DevBuf *b : bufs)
        if (b->p) e = hipFree(b->p);
    if (ctx->evt_ok)
        for (auto &ev : ctx->evt) if (ev) e = hipEventDestroy(ev);
    delete ctx;
}

const char *ns_last_error(const ns_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

static int set_ref_meta(ns_ctx *ctx, const uint64_t *chrom_off, uint32_t nchrom, const uint8_t *circular,
                        const char *names, uint64_t names_len) {
    if (!chrom_off || !nchrom || !circular || !names) return fail(ctx, NS_EINVAL, "reference metadata missing");
    free_pool(ctx->ref_allocs);
    ctx->nspecies = 0;                           // the species / expression views belong to the previous reference
    ctx->has_trx = false;
    ctx->has_ir = false;
    std::vector<uint32_t> noff(nchrom + 1);
    uint64_t p = 0;
    for (uint32_t c = 0; c < nchrom; ++c) {
        noff[c] = (uint32_t)p;
        while (p < names_len && names[p]) ++p;
        if (p >= names_len) return fail(ctx, NS_EINVAL, "names blob shorter than nchrom NUL-terminated strings");
        ++p;
    }
    noff[nchrom] = (uint32_t)p;
    int rc;
    if ((rc = upload(ctx, ctx->ref_allocs, chrom_off, (size_t)nchrom + 1, &ctx->ref.chrom_off))) return rc;
    if ((rc = upload(ctx, ctx->ref_allocs, circular, (size_t)nchrom, &ctx->ref.circular))) return rc;
    if ((rc = upload(ctx, ctx->ref_allocs, names, (size_t)p, &ctx->ref.names))) return rc;
    if ((rc = upload(ctx, ctx->ref_allocs, noff.data(), noff.size(), &ctx->ref.name_off))) return rc;
    ctx->ref.nchrom = nchrom;
    return NS_OK;
}

// The engine keeps its own normalised copy of the reference with NS_REF_PAD bytes of padding on both sides (the copy kernel
// issues unaligned 16-byte loads that may start a few bytes before / end a few bytes after a segment).
static int install_reference(ns_ctx *ctx, const void *src, bool src_on_device, uint64_t nbases, const uint64_t *chrom_off,
                             uint32_t nchrom, const uint8_t *circular, const char *names, uint64_t names_len) {
    HIPCHK(hipSetDevice(ctx->device));
    ctx->has_ref = false;
    if (chrom_off && nchrom && chrom_off[nchrom] != nbases) return fail(ctx, NS_EINVAL, "chrom_off[nchrom] != nbases");
    if (ctx->ref_bases_owned) { HIPCHK(hipFree(ctx->ref_bases_owned)); ctx->ref_bases_owned = nullptr; }
    HIPCHK(hipMalloc(&ctx->ref_bases_owned, nbases + 2 * NS_REF_PAD));
    uint8_t *data = static_cast<uint8_t *>(ctx->ref_bases_owned) + NS_REF_PAD;
    HIPCHK(hipMemsetAsync(ctx->ref_bases_owned, 'A', NS_REF_PAD, ctx->stream));
    HIPCHK(hipMemsetAsync(data + nbases, 'A', NS_REF_PAD, ctx->stream));
    HIPCHK(hipMemcpyAsync(data, src, nbases, src_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
    uint64_t nthreads = (nbases + 15) / 16;
    k_normalise<<<dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, ctx->stream>>>(data, nbases);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->ref.bases = data;
    int rc = set_ref_meta(ctx, chrom_off, nchrom, circular, names, names_len);
    if (rc) return rc;
    ctx->ref_nbases = nbases;
    ctx->hp_bm_k = 0;                    // (the homopolymer bitmap of the last reference)
    ctx->has_ref = true;
    return NS_OK;
}

int ns_set_reference(ns_ctx *ctx, const uint8_t *bases, uint64_t nbases, const uint64_t *chrom_off, uint32_t nchrom,
                     const uint8_t *circular, const char *names, uint64_t names_len) {
    NS_NOT_ON_COMPANION(ctx);
    if (!ctx) return NS_EINVAL;
    if (!bases || !nbases) return fail(ctx, NS_EINVAL, "empty reference");
    return install_reference(ctx, bases, false, nbases, chrom_off, nchrom, circular, names, names_len);
}

int ns_set_reference_device(ns_ctx *ctx, const void *bases_dev, uint64_t nbases, const uint64_t *chrom_off,
                            uint32_t nchrom, const uint8_t *circular, const char *names, uint64_t names_len) {
    NS_NOT_ON_COMPANION(ctx);
    if (!ctx) return NS_EINVAL;
    if (!bases_dev || !nbases) return fail(ctx, NS_EINVAL, "empty reference");
    return install_reference(ctx, bases_dev, true, nbases, chrom_off, nchrom, circular, names, names_len);
}

int ns_load_model(ns_ctx *ctx, const ns_model_tables *t) {
    NS_NOT_ON_COMPANION(ctx);
    if (!ctx) return NS_EINVAL;
    if (!t || t->abi_version != NS_ABI_VERSION) return fail(ctx, NS_EINVAL, "ns_model_tables: wrong abi_version");
    HIPCHK(hipSetDevice(ctx->device));
    ctx->has_model = false;
    free_pool(ctx->model_allocs);
    DevModel &m = ctx->m;
    memset(&m, 0, sizeof m);
    m.flags = t->flags;
    int rc;
    auto &pool = ctx->model_allocs;
    if (t->flags & NS_MODEL_HAS_ERRORS) {
        if (!t->fm_nseg || !t->mm_nbins) return fail(ctx, NS_EINVAL, "empty ECDF tables");
        m.fm_nseg = t->fm_nseg; m.fm_vlo0 = t->fm_vlo0;
        if ((rc = upload(ctx, pool, t->fm_hi, t->fm_nseg, &m.fm_hi))) return rc;
        if ((rc = upload(ctx, pool, t->fm_vhi, t->fm_nseg, &m.fm_vhi))) return rc;
        m.mm_nbins = t->mm_nbins;
        if (!t->mm_seg_off) return fail(ctx, NS_EINVAL, "mm_seg_off is null");
        uint32_t nseg = t->mm_seg_off[t->mm_nbins];
        if ((rc = upload(ctx, pool, t->mm_bin_lo, t->mm_nbins, &m.mm_bin_lo))) return rc;
        if ((rc = upload(ctx, pool, t->mm_bin_hi, t->mm_nbins, &m.mm_bin_hi))) return rc;
        if ((rc = upload(ctx, pool, t->mm_seg_off, (size_t)t->mm_nbins + 1, &m.mm_seg_off))) return rc;
        if ((rc = upload(ctx, pool, t->mm_hi, nseg, &m.mm_hi))) return rc;
        if ((rc = upload(ctx, pool, t->mm_vhi, nseg, &m.mm_vhi))) return rc;
        if ((rc = upload(ctx, pool, t->mm_vlo0, t->mm_nbins, &m.mm_vlo0))) return rc;
        memcpy(m.trans, t->trans, sizeof m.trans);
        double mean_match_min = 1e300;
        for (uint32_t b = 0; b < t->mm_nbins; ++b) {     // mean match length per bin -> event capacity per base
            double mean = 0, plo = 0, vlo = t->mm_vlo0[b];
            for (uint32_t s = t->mm_seg_off[b]; s < t->mm_seg_off[b + 1]; ++s) {
                mean += (t->mm_hi[s] - plo) * 0.5 * (vlo + t->mm_vhi[s]);
                plo = t->mm_hi[s]; vlo = t->mm_vhi[s];
            }
            if (mean < mean_match_min) mean_match_min = mean;
        }
        for (int ty = 0; ty < 3; ++ty) {
            m.mix_w[ty] = t->mix_w[ty];
            for (int c = 0; c < 2; ++c) {
                if (!t->mix_n[ty][c]) return fail(ctx, NS_EINVAL, "empty run-length table");
                m.mix_n[ty][c] = t->mix_n[ty][c];
                if ((rc = upload(ctx, pool, t->mix_cdf[ty][c], t->mix_n[ty][c], &m.mix_cdf[ty][c]))) return rc;
            }
        }
        double rate = 1.0 / (mean_match_min > 0.5 ? mean_match_min + 0.5 : 1.0);
        ctx->cap_rate = rate * 1.5 > 2.0 ? 2.0 : rate * 1.5;
        // test knob (tests/test_gpu_parity.py): a planned rate below the model's forces the event-capacity overflow and its re-plan
        if (const char *d = getenv("NS_CAP_RATE_SCALE")) { const double f = atof(d); if (f > 0.0) ctx->cap_rate *= f; }

        // ---- pack the chain tables into one blob of 8-byte words (its first part is copied to LDS by k_chain): ns_pack.h ----
        std::vector<uint64_t> blob;
        ChainTab &ct = m.ct;
        bool whole = true;
        uint32_t force_bits = 0;
        if (const char *d = getenv("NS_TAIL_BITS")) force_bits = (uint32_t)atoi(d) & 31u;
        ns_pack_chain_tables(t, nseg, ct, blob, whole, force_bits);
        if ((rc = upload(ctx, pool, blob.data(), blob.size(), &m.chain_blob))) return rc;
        ctx->lds_bytes = (size_t)ct.n_words_lds * 8;
        // (the packer sizes the image — hot prefixes of the match-length columns — for 24 / 32 / 45 KB: five / four / three workgroups of 256
        // threads per CU next to their 8 KB of event staging; a model whose image is larger still keeps its tables in global memory)
        ctx->lds_tables = whole && ctx->lds_bytes <= 44 * 1024;
        // workgroup size of the thread-per-read chain: 256 threads while five workgroups (image + 8 KB of staging) fit a CU's 160 KB, else
        // NS_CHAIN_BLOCK_BIG threads on one image while image + staging stays inside the 64 KB a launch may ask for
        ctx->chain_block = (ctx->lds_bytes + 8192 <= 32768 || ctx->lds_bytes + NS_CHAIN_BLOCK_BIG * 32u > 65536) ? NS_CHAIN_BLOCK : NS_CHAIN_BLOCK_BIG;
        if (const char *d = getenv("NS_CHAIN_BLOCK")) { const int v = atoi(d); if (v >= 64 && v <= NS_CHAIN_BLOCK_BIG && v % 64 == 0) ctx->chain_block = (uint32_t)v; }
        double vmax = 0;
        for (uint32_t k2 = 0; k2 < nseg; ++k2) if (t->mm_vhi[k2] > vmax) vmax = t->mm_vhi[k2];
        ctx->coop_ok = t->mm_nbins <= COOP_MAX_BINS && vmax < 65535.0;   // the cooperative chain keeps match lengths in 16 bits
    }
    for (int k = 0; k < NS_KDE_COUNT; ++k) {
        m.kde[k].n = t->kde[k].n; m.kde[k].bw = t->kde[k].bw;
        if ((rc = upload(ctx, pool, t->kde[k].data, (size_t)t->kde[k].n, &m.kde[k].data))) return rc;
    }
    m.strandness_rate = t->strandness_rate;
    if (t->flags & NS_MODEL_HAS_CHIMERIC) {
        m.nseg_n = t->nseg_n;
        if ((rc = upload(ctx, pool, t->nseg_cdf, t->nseg_n, &m.nseg_cdf))) return rc;
    }
    if (t->flags & NS_MODEL_HAS_QUALS) {
        if ((rc = upload(ctx, pool, &t->qual_thr[0][0], (size_t)NS_Q_COUNT * NS_QUAL_LEVELS, &m.qual_thr))) return rc;
        std::vector<uint16_t> lut((size_t)NS_Q_COUNT * 1024);
        for (int c = 0; c < NS_Q_COUNT; ++c)
            for (uint32_t b = 0; b < 1024; ++b) {
                uint32_t q = 0;
                while (q < NS_QUAL_LEVELS - 1 && t->qual_thr[c][q] <= 64u * b) ++q;      // thresholds at or below the bucket start
                uint32_t inside = 0, sub = 64;                                             // thresholds in (64 b, 64 b + 63]
                for (uint32_t j = q; j < NS_QUAL_LEVELS - 1 && t->qual_thr[c][j] <= 64u * b + 63u; ++j) { if (!inside) sub = t->qual_thr[c][j] - 64u * b; ++inside; }
                // q << 7 | (128 - sub): adding h & 63 carries into the count exactly when h & 63 >= sub (see qual_value_lut)
                lut[(size_t)c * 1024 + b] = (uint16_t)(q << 7 | (128u - sub) | (inside > 1 ? 0x8000u : 0u));
            }
        if ((rc = upload(ctx, pool, lut.data(), lut.size(), &m.qual_lut))) return rc;
    }
    memcpy(m.hp, t->hp, sizeof m.hp);
    m.hp_mis_rate = t->hp_mis_rate;
    if (t->flags & NS_MODEL_HAS_KDE2D) {
        if (!t->kde2d_n || !(t->kde2d_bw > 0)) return fail(ctx, NS_EINVAL, "empty 2-D KDE");
        for (uint64_t i = 1; i < t->kde2d_n; ++i)
            if (t->kde2d_x[i] < t->kde2d_x[i - 1]) return fail(ctx, NS_EINVAL, "kde2d_x must be sorted ascending");
        if ((rc = upload(ctx, pool, t->kde2d_x, (size_t)t->kde2d_n, &m.kde2d_x)) || (rc = upload(ctx, pool, t->kde2d_y, (size_t)t->kde2d_n, &m.kde2d_y))) return rc;
        m.kde2d_n = t->kde2d_n; m.kde2d_bw = t->kde2d_bw;
    }
    ctx->has_model = true;
    return NS_OK;
}

#define NS_OVER_MASK ((1ull << 40) - 1)          // stats[0]: capacity overflows in the low bits, attempts dropped for the event-record range above
static int scan_u64(ns_ctx *ctx, const uint64_t *in, uint64_t *out, size_t n) {
    size_t tmp = 0;
    HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp, in, out, (int)n, ctx->stream));
    int rc = ensure(ctx, ctx->scan_tmp, tmp);
    if (rc) return rc;
    HIPCHK(hipcub::DeviceScan::ExclusiveSum(ctx->scan_tmp.p, tmp, in, out, (int)n, ctx->stream));
    return NS_OK;
}
static int scan_u32(ns_ctx *ctx, const uint32_t *in, uint32_t *out, size_t n) {
    size_t tmp = 0;
    HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp, in, out, (int)n, ctx->stream));
    int rc = ensure(ctx, ctx->scan_tmp, tmp);
    if (rc) return rc;
    HIPCHK(hipcub::DeviceScan::ExclusiveSum(ctx->scan_tmp.p, tmp, in, out, (int)n, ctx->stream));
    return NS_OK;
}


// ---------------------------------------------------------------------------------------------------------
// metagenome (src/simulator.py:758-811, 814-1040)
// ---------------------------------------------------------------------------------------------------------
// copy phase, slow tiles, payload: the three kernels that write the sequence (and quality) lines of a batch
static int launch_materialise(ns_ctx *ctx, const GenArgs &A, size_t n, bool fastq, uint64_t event_slots, hipEvent_t names_done = nullptr,
                              const uint32_t *order = nullptr, int mode = MAT_REF, uint64_t total_bases = 0) {
    hipStream_t st = ctx->stream;
    if (A.prm.kind == NS_KIND_UNALIGNED) {
        // work list: stretches per read + scan (list_b / list_c are free once the passes are done); the grid is bounded from the batch's
        // total: every read adds at most one partial stretch.  (Round 5, next to an aligned worker call — ns_generate_step — the unaligned
        // call ends ~0.75 ms after the aligned one: its dense kernel gets no wavefront slots while the record kernel's grid still has
        // workgroups to start.  Neither fewer launches in this tail — plan + both scans as ONE single-workgroup kernel: 10.0-10.2 against
        // 9.7 ms per step — nor the aligned record kernel as two launches on two streams (80 % + 20 %: 9.75-9.82 against 9.81-10.1, noise)
        // nor stream priorities moved it: profiles/r05/ab_step_gate.log.)
        uint32_t *cnt = (uint32_t *)ctx->list_b.p, *seg_off = (uint32_t *)ctx->list_c.p;
        k_dense_plan<<<dim3((unsigned)((n + 1 + 255) / 256)), dim3(256), 0, st>>>(A, cnt);
        HIPCHK(hipGetLastError());
        if (int rc = scan_u32(ctx, cnt, seg_off, n + 1)) return rc;
        const uint64_t bound = total_bases / NS_DENSE_SEG + n + 1;
        if (bound > 0x7fffffffull) return fail(ctx, NS_EINVAL, "unaligned batch too large for one launch of the record kernel (split it)");
        if (names_done) HIPCHK(hipStreamWaitEvent(st, names_done, 0));
        // FASTQ: the bases here, the quality lines in k_qualities (one class for the whole read, S:1521: no class words) — drawn inside the
        // dense kernel they came through the per-value look-up in global memory: 1.25 against 0.7 ms per 50 000 reads
        k_materialise_dense<false><<<dim3((unsigned)bound), dim3(64), 0, st>>>(A, seg_off);
        HIPCHK(hipGetLastError());
        if (fastq) {
            k_qualities<false><<<dim3((unsigned)((n + NS_MATQ_WAVES - 1) / NS_MATQ_WAVES)), dim3(64 * NS_MATQ_WAVES), 0, st>>>(A, nullptr);
            HIPCHK(hipGetLastError());
        }
        return NS_OK;
    }
    (void)event_slots;                        // (the letter words are drawn in the tile prologue: event_word, ns_materialise.h)
    if (names_done) HIPCHK(hipStreamWaitEvent(st, names_done, 0));
    for (int round = 0;; ++round) {
        size_t cap = ctx->slow_q.cap >= 16 + sizeof(SlowTile) ? (ctx->slow_q.cap - 16) / sizeof(SlowTile) : 0;
        if (cap < n / 4 + 4096) {
            int rc = ensure(ctx, ctx->slow_q, 16 + (n / 4 + 4096) * sizeof(SlowTile));
            if (rc) return rc;
            cap = (ctx->slow_q.cap - 16) / sizeof(SlowTile);
            ctx->sq_zeroed = false;                 // a NEW buffer: k_stats_fold zeroed the counter of the one just freed.  (Until the last day of round 6 the
        }                                           // launch went ahead on whatever the new allocation held — a batch larger than all before it, on recycled device
                                                    // memory, queued `garbage` tiles: a memory fault in the generic kernel; scripts/parity_chimeric_big.py found it)
        SlowQueue sq;
        sq.count = (uint32_t *)ctx->slow_q.p; sq.items = (SlowTile *)((uint8_t *)ctx->slow_q.p + 16);
        sq.cap = (uint32_t)(cap > 0xffffffffull ? 0xffffffffull : cap);
        if (!(ctx->sq_zeroed && round == 0)) HIPCHK(hipMemsetAsync(sq.count, 0, 4, st));
        ctx->sq_zeroed = false;
        const uint32_t *wd = nullptr;               // (MAT_HP_FINAL reads A.hp_wd; the other modes draw the letter words: event_word)
        if (ctx->dbg & 1024u) order = nullptr;          // (profiling: reads in index order)
        const dim3 grid_q((unsigned)((n + NS_MATQ_WAVES - 1) / NS_MATQ_WAVES)), blk_q(64 * NS_MATQ_WAVES), grid_1((unsigned)n), blk_1(64);
        const uint32_t *order_b = (ctx->dbg & 2048u) ? order : nullptr;      // (the record kernel: reads in index order)
        if (round == 0) HIPCHK(hipEventRecord(ctx->evt[12], st));            // the record kernel itself (ns_batch_info.ms_kernel[NS_K_RECORD_KERNEL])
        if (mode == MAT_REF) {
            if (fastq) k_materialise<true, MAT_REF><<<grid_1, blk_1, 0, st>>>(A, wd, ctx->dbg, sq, order_b);
            else k_materialise<false, MAT_REF><<<grid_1, blk_1, 0, st>>>(A, wd, ctx->dbg, sq, order_b);
        } else if (mode == MAT_HP_SCRATCH) {
            if (fastq) k_materialise<true, MAT_HP_SCRATCH><<<grid_1, blk_1, 0, st>>>(A, wd, ctx->dbg, sq, nullptr);
            else k_materialise<false, MAT_HP_SCRATCH><<<grid_1, blk_1, 0, st>>>(A, wd, ctx->dbg, sq, nullptr);
        } else {
            if (fastq) k_materialise<true, MAT_HP_FINAL><<<grid_1, blk_1, 0, st>>>(A, wd, ctx->dbg, sq, order_b);
            else k_materialise<false, MAT_HP_FINAL><<<grid_1, blk_1, 0, st>>>(A, wd, ctx->dbg, sq, nullptr);
        }
        HIPCHK(hipGetLastError());
        if (round == 0) { HIPCHK(hipEventRecord(ctx->evt[13], st)); ctx->rec_timed = true; }
        // FASTQ: the quality lines, from the class words the record kernel left (before the generic kernel below: a tile queued for it
        // has no class words, and its qualities are that kernel's)
        if (fastq && mode != MAT_HP_SCRATCH && round == 0) {
            if (mode == MAT_HP_FINAL) k_qualities<true><<<grid_q, blk_q, 0, st>>>(A, order);
            else k_qualities<false><<<grid_q, blk_q, 0, st>>>(A, order);
            HIPCHK(hipGetLastError());
        }
        uint32_t queued = 0;
        if (int rc2 = read_small(ctx, st, &queued, sq.count, 4)) return rc2;
        if (queued > sq.cap) {                   // more slow tiles than queue slots (tiny circular genomes): grow and redo
            if (round >= 2) return fail(ctx, NS_ENOMEM, "slow-tile queue overflow");
            int rc = ensure(ctx, ctx->slow_q, 16 + ((size_t)queued + 4096) * sizeof(SlowTile));
            if (rc) return rc;
            continue;
        }
        if (queued) {
            const unsigned grid = queued < 16384u ? queued : 16384u;
            if (mode == MAT_HP_FINAL) {          // >= 64 homopolymer edits at one output offset (adjacent runs all re-sampled to nothing)
                if (fastq) k_materialise_slow_hpf<true><<<dim3(grid), dim3(64), 0, st>>>(A, sq);
                else k_materialise_slow_hpf<false><<<dim3(grid), dim3(64), 0, st>>>(A, sq);
            } else
            if (mode == MAT_HP_SCRATCH) {
                if (fastq) k_materialise_slow<true, true><<<dim3(grid), dim3(64), 0, st>>>(A, sq);
                else k_materialise_slow<false, true><<<dim3(grid), dim3(64), 0, st>>>(A, sq);
            } else if (fastq) k_materialise_slow<true, false><<<dim3(grid), dim3(64), 0, st>>>(A, sq);
            else k_materialise_slow<false, false><<<dim3(grid), dim3(64), 0, st>>>(A, sq);
            HIPCHK(hipGetLastError());
        }
        return NS_OK;
    }
}

// -k stage 1 on the reads of `A` (A.prm.n_reads of them): filter the events inside homopolymers (S:1920-1947), write the pieces before
// mutate_homo to the scratch buffer, turn mutate_homo (S:618-706) into an edit list per piece + final lengths, and apply the final
// length check (stats[5] = reads that failed it)
static int hp_stage1(ns_ctx *ctx, const ns_params *prm, GenArgs &A, size_t n, uint64_t tot_pieces, uint64_t event_slots,
                     unsigned long long *stats, double *ms_hp) {
    hipStream_t st = ctx->stream;
    const dim3 blk(256), grid_t((unsigned)((n + 1 + 255) / 256));
    int rc;
    float ms = 0;
    HIPCHK(hipEventRecord(ctx->evt[9], st));
    if ((rc = ensure(ctx, ctx->hp_len, (size_t)tot_pieces * 4 + 64)) || (rc = ensure(ctx, ctx->hp_nev, (size_t)tot_pieces * 4 + 64)) ||
        (rc = ensure(ctx, ctx->hp_nrun, (size_t)tot_pieces * 4 + 64))) return rc;
    A.hp_len = (uint32_t *)ctx->hp_len.p; A.hp_nev = (uint32_t *)ctx->hp_nev.p;
    if ((rc = ensure(ctx, ctx->hp_pcnt, (n + 1) * 4)) || (rc = ensure(ctx, ctx->hp_pord, (n + 1) * 4))) return rc;
    A.hp_pcnt = (uint32_t *)ctx->hp_pcnt.p; A.hp_pord = (const uint32_t *)ctx->hp_pord.p;
    A.hp_bm = nullptr;
    if (prm->kmer_bias >= 2 && prm->kmer_bias <= 16 && !getenv("NS_NO_HP_BITMAP")) {
        if (ctx->hp_bm_k != prm->kmer_bias) {           // once per (reference, k)
            const uint64_t nb = ctx->ref_nbases, nthreads = (nb + 3) / 4;
            if ((rc = ensure(ctx, ctx->hp_bm, (size_t)nthreads + 64))) return rc;
            HIPCHK(hipMemsetAsync((uint8_t *)ctx->hp_bm.p + nthreads, 0, 64, st));       // (8-byte loads may run past the last base)
            k_hp_bitmap<<<dim3((unsigned)((nthreads + 255) / 256)), blk, 0, st>>>(ctx->ref.bases, nb, prm->kmer_bias, (uint8_t *)ctx->hp_bm.p);
            HIPCHK(hipGetLastError());
            ctx->hp_bm_k = prm->kmer_bias;
        }
        A.hp_bm = (const uint8_t *)ctx->hp_bm.p;
    }
    k_hp_filter_w<<<dim3((unsigned)((n + NS_WPB) / NS_WPB)), dim3(64 * NS_WPB), 0, st>>>(A);
    HIPCHK(hipGetLastError());
    if ((rc = scan_u64(ctx, A.scr_len, A.scr_off, n + 1)) || (rc = scan_u32(ctx, A.hp_pcnt, (uint32_t *)ctx->hp_pord.p, n + 1))) return rc;
    uint64_t scr_bytes = 0;
    if ((rc = read_small(ctx, st, &scr_bytes, A.scr_off + n, 8))) return rc;
    // (the second record pass reads the scratch pieces with unaligned 16-byte loads that may start before / end behind a piece)
    if ((rc = ensure(ctx, ctx->scr, (size_t)scr_bytes + 2 * NS_REF_PAD + 64))) return rc;
    A.scr = (uint8_t *)ctx->scr.p + NS_REF_PAD;
    if ((rc = launch_materialise(ctx, A, n, prm->fastq != 0, event_slots, nullptr, nullptr, MAT_HP_SCRATCH))) return rc;
    if (ctx->hp_cap_k != prm->kmer_bias) {      // event capacity per scratch byte: 8x the density of runs >= k in a random sequence
        double rate = 6.0;
        for (uint32_t j = 1; j < prm->kmer_bias && rate > 1e-6; ++j) rate *= 0.25;
        uint32_t sh = 0;
        while (sh < 16 && rate * (double)(2u << sh) <= 1.0) ++sh;
        ctx->hp_shift = sh; ctx->hp_pad = 64; ctx->hp_cap_k = prm->kmer_bias;
        // test knob: NS_HP_CAP_SHIFT=s plans 2^-s of that capacity and one slot of slack per piece, so that the stage's own overflow path runs
        if (const char *d = getenv("NS_HP_CAP_SHIFT")) { const int s = atoi(d); if (s > 0) { ctx->hp_shift = std::min<uint32_t>(16u, sh + (uint32_t)s); ctx->hp_pad = 1; } }
    }
    for (int retry = 0;; ++retry) {
        A.hp_shift = ctx->hp_shift; A.hp_pad = ctx->hp_pad;
        const size_t slots = (size_t)(scr_bytes >> A.hp_shift) + (size_t)A.hp_pad * (tot_pieces + 1) + 64;
        if ((rc = ensure(ctx, ctx->hp_ev, slots * sizeof(ns_event))) || (rc = ensure(ctx, ctx->hp_wd, slots * 4)) || (rc = ensure(ctx, ctx->hp_runs, slots * 8))) return rc;
        A.hp_ev = (ns_event *)ctx->hp_ev.p; A.hp_wd = (uint32_t *)ctx->hp_wd.p;
        if (!A.meta || A.key_pos) HIPCHK(hipMemsetAsync((unsigned long long *)ctx->stats.p + 1, 0, sizeof(unsigned long long), st));   // (kept across metagenome passes)
        HIPCHK(hipMemsetAsync((unsigned long long *)ctx->stats.p + 5, 0, sizeof(unsigned long long), st));
        HIPCHK(hipMemsetAsync((unsigned long long *)ctx->stats.p + 7, 0, sizeof(unsigned long long), st));
        k_hp_scan<<<dim3((unsigned)((n + NS_WPB - 1) / NS_WPB)), dim3(64 * NS_WPB), 0, st>>>(A, (uint2 *)ctx->hp_runs.p, (uint32_t *)ctx->hp_nrun.p);
        k_hp_drain<<<dim3((unsigned)((n + NS_WPB - 1) / NS_WPB)), dim3(64 * NS_WPB), 0, st>>>(A, (const uint2 *)ctx->hp_runs.p, (const uint32_t *)ctx->hp_nrun.p, A.l_cap);
        k_hp_finalize<<<grid_t, blk, 0, st>>>(A, A.l_cap);
        k_sum_u64<<<dim3((unsigned)std::min<size_t>(512, (n + 255) / 256)), blk, 0, st>>>(A.scr_len, n, (unsigned long long *)ctx->stats.p + 1);
        HIPCHK(hipGetLastError());
        if ((rc = read_small(ctx, st, stats, ctx->stats.p, 8 * sizeof(unsigned long long)))) return rc;
        if (!stats[7]) break;
        // more homopolymer edits per base than planned (low-complexity reference): nothing was finalised; again with twice the capacity
        if (retry >= 12) return fail(ctx, NS_ENOMEM, "-k: event capacity overflow persists");
        if (ctx->hp_shift) --ctx->hp_shift;
        ctx->hp_pad *= 2;
    }
    HIPCHK(hipEventRecord(ctx->evt[10], st));
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipEventElapsedTime(&ms, ctx->evt[9], ctx->evt[10]));
    *ms_hp += ms;
    return NS_OK;
}

int ns_set_transcriptome(ns_ctx *ctx, uint32_t n_expr, const uint32_t *expr_chrom, const double *expr_cum, const uint8_t *polya,
                         double polya_scale) {
    NS_NOT_ON_COMPANION(ctx);
    if (!ctx) return NS_EINVAL;
    if (!ctx->has_ref) return fail(ctx, NS_ESTATE, "ns_set_transcriptome before ns_set_reference");
    if (!n_expr || !expr_chrom || !expr_cum) return fail(ctx, NS_EINVAL, "empty expression table");
    for (uint32_t i = 0; i < n_expr; ++i) {
        if (expr_chrom[i] >= ctx->ref.nchrom) return fail(ctx, NS_EINVAL, "expression table points outside the reference");
        if (i && expr_cum[i] < expr_cum[i - 1]) return fail(ctx, NS_EINVAL, "expr_cum must be non-decreasing");
    }
    if (!(expr_cum[n_expr - 1] > 0)) return fail(ctx, NS_EINVAL, "no expression weight");
    HIPCHK(hipSetDevice(ctx->device));
    int rc;
    if ((rc = ensure(ctx, ctx->trx_chrom, (size_t)n_expr * 4)) || (rc = ensure(ctx, ctx->trx_cum, (size_t)n_expr * 8))) return rc;
    HIPCHK(hipMemcpy(ctx->trx_chrom.p, expr_chrom, (size_t)n_expr * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ctx->trx_cum.p, expr_cum, (size_t)n_expr * 8, hipMemcpyHostToDevice));
    ctx->tx.n_expr = n_expr; ctx->tx.expr_chrom = (const uint32_t *)ctx->trx_chrom.p; ctx->tx.expr_cum = (const double *)ctx->trx_cum.p;
    ctx->tx.polya = nullptr; ctx->tx.polya_scale = polya_scale;
    if (polya) {
        if ((rc = ensure(ctx, ctx->trx_polya, (size_t)ctx->ref.nchrom))) return rc;
        HIPCHK(hipMemcpy(ctx->trx_polya.p, polya, (size_t)ctx->ref.nchrom, hipMemcpyHostToDevice));
        ctx->tx.polya = (const uint8_t *)ctx->trx_polya.p;
    }
    ctx->has_trx = true;
    return NS_OK;
}

int ns_set_intron_retention(ns_ctx *ctx, const ns_ir_tables *t) {
    NS_NOT_ON_COMPANION(ctx);
    if (!ctx) return NS_EINVAL;
    HIPCHK(hipSetDevice(ctx->device));
    ctx->has_ir = false;
    free_pool(ctx->ir_allocs);
    ctx->ir = DevIr{};
    if (!t) return NS_OK;
    if (!ctx->has_ref) return fail(ctx, NS_ESTATE, "ns_set_intron_retention before ns_set_reference");
    const uint32_t ntr = ctx->ref.nchrom;
    if (!t->genome || !t->genome_off || !t->n_gchrom || !t->item_off) return fail(ctx, NS_EINVAL, "intron retention tables missing");
    if (t->item_off[0] != 0 || t->item_off[ntr] != t->n_items) return fail(ctx, NS_EINVAL, "item_off does not cover n_items");
    for (uint32_t c = 0; c < ntr; ++c)
        if (t->item_off[c] > t->item_off[c + 1]) return fail(ctx, NS_EINVAL, "item_off not ascending");
    for (uint32_t i = 0; i < t->n_items; ++i) {
        if (t->item_type[i] > NS_IR_INTRON) return fail(ctx, NS_EINVAL, "bad item type");
        const uint32_t c = t->item_chrom[i];
        if (c == NS_IR_NO_CHROM) continue;
        if (c >= t->n_gchrom) return fail(ctx, NS_EINVAL, "item chromosome out of range");
        if ((uint64_t)t->item_start[i] + t->item_len[i] > t->genome_off[c + 1] - t->genome_off[c])
            return fail(ctx, NS_EINVAL, "item reaches beyond its chromosome");
    }
    for (int k = 0; k < 3; ++k)
        if (!(t->p_no_ir[k] >= 0 && t->p_ir[k] >= 0 && t->p_no_ir[k] + t->p_ir[k] <= 1.0 + 1e-9))
            return fail(ctx, NS_EINVAL, "IR Markov model rows must be probabilities");
    int rc;
    DevIr d{};
    const uint64_t glen = t->genome_off[t->n_gchrom];
    if ((rc = upload(ctx, ctx->ir_allocs, t->genome, (size_t)glen, &d.genome)) ||
        (rc = upload(ctx, ctx->ir_allocs, t->genome_off, (size_t)t->n_gchrom + 1, &d.genome_off)) ||
        (rc = upload(ctx, ctx->ir_allocs, t->item_off, (size_t)ntr + 1, &d.item_off)))
        return rc;
    if (t->n_items &&
        ((rc = upload(ctx, ctx->ir_allocs, t->item_type, (size_t)t->n_items, &d.item_type)) ||
         (rc = upload(ctx, ctx->ir_allocs, t->item_minus, (size_t)t->n_items, &d.item_minus)) ||
         (rc = upload(ctx, ctx->ir_allocs, t->item_chrom, (size_t)t->n_items, &d.item_chrom)) ||
         (rc = upload(ctx, ctx->ir_allocs, t->item_start, (size_t)t->n_items, &d.item_start)) ||
         (rc = upload(ctx, ctx->ir_allocs, t->item_len, (size_t)t->n_items, &d.item_len))))
        return rc;
    for (int k = 0; k < 3; ++k) { d.p_no_ir[k] = t->p_no_ir[k]; d.p_ir[k] = t->p_ir[k]; }
    ctx->ir = d;
    ctx->has_ir = true;
    return NS_OK;
}

int ns_set_species(ns_ctx *ctx, uint32_t nspecies, const uint32_t *species_chrom_off) {
    NS_NOT_ON_COMPANION(ctx);
    if (!ctx) return NS_EINVAL;
    if (!ctx->has_ref) return fail(ctx, NS_ESTATE, "ns_set_species before ns_set_reference");
    if (!nspecies || nspecies > 65535u || !species_chrom_off) return fail(ctx, NS_EINVAL, "bad species table");
    if (species_chrom_off[0] != 0 || species_chrom_off[nspecies] != ctx->ref.nchrom) return fail(ctx, NS_EINVAL, "species_chrom_off does not cover the reference");
    for (uint32_t s = 0; s < nspecies; ++s)
        if (species_chrom_off[s + 1] <= species_chrom_off[s]) return fail(ctx, NS_EINVAL, "species without chromosomes");
    HIPCHK(hipSetDevice(ctx->device));
    int rc = ensure(ctx, ctx->species_chrom_off, ((size_t)nspecies + 1) * 4);
    if (rc) return rc;
    HIPCHK(hipMemcpy(ctx->species_chrom_off.p, species_chrom_off, ((size_t)nspecies + 1) * 4, hipMemcpyHostToDevice));
    ctx->nspecies = nspecies;
    ctx->has_abun = ctx->has_inflated = false;
    return NS_OK;
}

int ns_set_abundance(ns_ctx *ctx, const double *abun, const double *abun_inflated) {
    NS_NOT_ON_COMPANION(ctx);
    if (!ctx) return NS_EINVAL;
    if (!ctx->nspecies) return fail(ctx, NS_ESTATE, "ns_set_abundance before ns_set_species");
    if (!abun) return fail(ctx, NS_EINVAL, "null abundance table");
    ctx->abun.assign(abun, abun + ctx->nspecies);
    ctx->has_abun = true;
    ctx->has_inflated = abun_inflated != nullptr;
    if (abun_inflated) ctx->abun_inflated.assign(abun_inflated, abun_inflated + ctx->nspecies);
    return NS_OK;
}

int ns_species_bases(ns_ctx *ctx, double *out) {
    if (!ctx) return NS_EINVAL;
    if (!out) return fail(ctx, NS_EINVAL, "null destination");
    if (!ctx->has_batch || ctx->last_species_bases.size() != ctx->nspecies) return fail(ctx, NS_ESTATE, "no metagenome batch");
    for (uint32_t s = 0; s < ctx->nspecies; ++s) out[s] = ctx->last_species_bases[s];
    return NS_OK;
}

// assign_species (S:758-811): the species of every segment of a pass, by greedy quota.  lens: the filtered length list of the
// pass (in draw order); on return the list in assignment order (chimeric segments first, the rest descending).  Draws keyed by
// the batch: Philox(ST_SPECIES, attempt = pass, idx = segment pointer): word 0 = random.choice, word 1 = random.uniform(0, 100).
// assign_species (S:758-811): the species of every segment of a pass, by greedy quota — a sequential walk, on the host.  `lens`: the
// filtered length list in assignment order (chimeric segments first in draw order, the rest descending — sorted on the device);
// `to_add`: sum(length_list) taken left to right over the list in draw order; `words`: the draws of every segment pointer.
// `hist[v]`: how many of the reads still missing have v segments — the reads are taken in descending order of v (S:760: a descending sort of
// num_segment).  Returns the segments assigned; `reads_done`: the reads whose segments were all assigned.
// The walk is sequential by definition (every assignment lowers a quota).  Chimeric reads (their lengths are in draw order) take the
// loop over all species as the reference writes it.  The single-segment reads — nine in ten — come by DESCENDING length (S:764-765), so
// "quota - len > 0" can only become true for a species as the walk goes on, and false only for the species just charged: the candidate
// list is kept up to date with O(1) work per read instead of being rebuilt from all species (same comparisons, same order of the
// candidates, same arithmetic).
static uint64_t assign_species_host(const ns_ctx *ctx, const double *lens, uint64_t n_len, double to_add, const uint2 *words,
                                    const uint64_t *hist, const std::vector<double> &cur_bases, uint16_t *species, uint64_t *reads_done) {
    const uint32_t ns = ctx->nspecies;
    *reads_done = 0;
    double have = 0, abun_total = 0;
    for (uint32_t s = 0; s < ns; ++s) { have += cur_bases[s]; abun_total += ctx->abun[s]; }
    const double all_bases = to_add + have;
    std::vector<double> quota(ns);
    for (uint32_t s = 0; s < ns; ++s) quota[s] = all_bases * ctx->abun[s] / abun_total - cur_bases[s];   // S:772-775
    std::vector<uint32_t> cand(ns);
    uint64_t ptr = 0;
    uint32_t prev = 0;
    auto fitting = [&](double len, int skip) {              // species whose quota still holds `len` (S:785-788: else any with quota left)
        uint32_t c = 0;
        for (uint32_t s = 0; s < ns; ++s) if (quota[s] - len > 0 && (int)s != skip) cand[c++] = s;
        return c;
    };
    auto any_left = [&]() { uint32_t c = 0; for (uint32_t s = 0; s < ns; ++s) if (quota[s] > 0) cand[c++] = s; return c; };
    for (int32_t seg = (int32_t)NS_MAX_SEG; seg >= 2; --seg) {                  // ---- chimeric reads
        for (uint64_t r = 0; r < hist[seg]; ++r) {
            if (ptr + (uint64_t)seg > n_len) return ptr;                         // S:781-782
            for (int32_t k = 0; k < seg; ++k) {
                const double len = lens[ptr];
                const uint2 w = words[ptr];                 // Philox(batch, ST_SPECIES, attempt = pass, idx = ptr): .x choice, .y uniform(0, 100)
                auto choose = [&](uint32_t c) { return cand[(uint32_t)(((uint64_t)w.x * c) >> 32)]; };
                uint32_t sp = 0, c;
                bool fresh = k == 0;
                if (!fresh) {                                                    // S:791-803: stay with the previous species?
                    c = fitting(len, (int)prev);
                    const double pct = 100.0 * u32_to_p(w.y);
                    if (pct <= ctx->abun_inflated[prev] && quota[prev] > 0) sp = prev;
                    else if (pct > ctx->abun_inflated[prev] && c > 0) sp = choose(c);
                    else fresh = true;
                }
                if (fresh) {
                    c = fitting(len, -1);
                    if (!c) c = any_left();
                    if (!c) return ptr;
                    sp = choose(c);
                }
                species[ptr] = (uint16_t)sp;
                quota[sp] -= len;
                prev = sp;
                ++ptr;
            }
            ++*reads_done;
        }
    }
    // ---- single-segment reads, lengths descending: fit = {s : quota[s] - len > 0} in species order, marg = {s : 0 < quota[s], not fitting}
    std::vector<uint32_t> fit(ns), marg(ns);
    uint32_t n_fit = 0, n_marg = 0;
    const uint64_t n1 = hist[1];
    if (n1 && ptr < n_len) {
        const double len0 = lens[ptr];
        for (uint32_t s = 0; s < ns; ++s) { if (quota[s] - len0 > 0) fit[n_fit++] = s; else if (quota[s] > 0) marg[n_marg++] = s; }
    }
    uint32_t last = 0xffffffffu;                            // the species charged by the previous read (the only one that can leave `fit`)
    for (uint64_t r = 0; r < n1; ++r) {
        if (ptr + 1 > n_len) return ptr;                                         // S:781-782
        const double len = lens[ptr];
        const uint2 w = words[ptr];
        if (last != 0xffffffffu && !(quota[last] - len > 0)) {                   // ... it no longer holds this length
            uint32_t i = 0;
            while (i < n_fit && fit[i] != last) ++i;
            if (i < n_fit) { for (; i + 1 < n_fit; ++i) fit[i] = fit[i + 1]; --n_fit; if (quota[last] > 0) marg[n_marg++] = last; }
        }
        for (uint32_t i = 0; i < n_marg;) {                                      // a species with little quota left fits the shorter reads again
            const uint32_t s = marg[i];
            if (quota[s] - len > 0) {
                uint32_t j = n_fit++;
                while (j > 0 && fit[j - 1] > s) { fit[j] = fit[j - 1]; --j; }
                fit[j] = s;
                marg[i] = marg[--n_marg];
            } else ++i;
        }
        uint32_t sp;
        if (n_fit) sp = fit[(uint32_t)(((uint64_t)w.x * n_fit) >> 32)];
        else {                                                                   // S:787-788: any species with quota left, in species order
            const uint32_t c = any_left();
            if (!c) return ptr;
            sp = cand[(uint32_t)(((uint64_t)w.x * c) >> 32)];
        }
        species[ptr] = (uint16_t)sp;
        quota[sp] -= len;
        if (!(quota[sp] > 0)) {                                                  // used up: out of both lists
            for (uint32_t i = 0; i < n_marg; ++i) if (marg[i] == sp) { marg[i] = marg[--n_marg]; break; }
            uint32_t i = 0;
            while (i < n_fit && fit[i] != sp) ++i;
            if (i < n_fit) { for (; i + 1 < n_fit; ++i) fit[i] = fit[i + 1]; --n_fit; }
            last = 0xffffffffu;
        } else last = sp;
        prev = sp;
        ++ptr;
        ++*reads_done;
    }
    return ptr;
}

// histogram of num_segment over the reads [lo, n) of a metagenome batch (n_pieces = 2 * num_segment - 1): what the passes need of
// remaining_segments (S:1034) — the reads are taken by descending segment count, so the counts per value say everything
__global__ void __launch_bounds__(256) k_meta_hist(const uint32_t *__restrict__ n_pieces, uint64_t lo, uint64_t n, unsigned long long *__restrict__ hist) {
    __shared__ uint32_t h[NS_MAX_SEG + 1];
    if (threadIdx.x <= NS_MAX_SEG) h[threadIdx.x] = 0;
    __syncthreads();
    for (uint64_t i = lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        atomicAdd(&h[min((n_pieces[i] + 1u) / 2u, (uint32_t)NS_MAX_SEG)], 1u);
    __syncthreads();
    if (threadIdx.x <= NS_MAX_SEG && h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}

// positions of a pass in assignment order -> first segment / first piece of the read: the reads are sorted by descending segment count,
// so both are closed forms of the histogram (S:862-865)
struct MetaHist { uint32_t cnt[NS_MAX_SEG + 1]; };
__global__ void __launch_bounds__(256) k_meta_layout(MetaHist H, uint32_t np, uint32_t *__restrict__ segptr, uint32_t *__restrict__ pieceoff) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > np) return;
    uint32_t start = 0, sbase = 0, pbase = 0;
    for (int v = (int)NS_MAX_SEG; v >= 1; --v) {
        const uint32_t c = H.cnt[v];
        if (i < start + c) { segptr[i] = sbase + (i - start) * (uint32_t)v; pieceoff[i] = pbase + (i - start) * (2u * (uint32_t)v - 1u); return; }
        start += c; sbase += c * (uint32_t)v; pbase += c * (2u * (uint32_t)v - 1u);
    }
    segptr[i] = sbase; pieceoff[i] = pbase;                 // i == the number of reads: the totals
}

// the passes of one metagenome worker: every pass draws fresh lengths for the reads still missing, assigns species and tries
// each read once; accepted reads take consecutive numbers
static int meta_passes(ns_ctx *ctx, const ns_params *prm, ns_batch_info *info, GenArgs &A, uint64_t &tot_pieces, uint64_t &tot_cap,
                       unsigned long long *stats) {
    const size_t n = (size_t)prm->n_reads;
    const uint32_t ns = ctx->nspecies;
    hipStream_t st = ctx->stream;
    const dim3 blk(256);
    int rc;
    float ms = 0;
    HIPCHK(hipMemsetAsync(ctx->stats.p, 0, NS_STATS_BYTES, st));
    HIPCHK(hipEventRecord(ctx->evt[1], st));
    k_nseg<<<dim3((unsigned)((n + 1 + 255) / 256)), blk, 0, st>>>(A);       // num_segment (S:825-828); zeroes the scan sentinels
    HIPCHK(hipGetLastError());
    const bool trace = getenv("NS_META_TRACE") != nullptr;      // host-side section timing of the passes (stderr)
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto lap = [&](const char *what, std::chrono::steady_clock::time_point &t) {
        if (!trace) return;
        auto t2 = now();
        fprintf(stderr, "[meta] %-22s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t2 - t).count());
        t = t2;
    };
    auto tt = now();
    // num_segment (k_nseg) stays on the device: the passes need its histogram over the reads still missing, 65 counters per pass
    if ((rc = ensure(ctx, ctx->meta_num, 16 + (NS_MAX_SEG + 1) * 8))) return rc;
    unsigned long long *d_hist = (unsigned long long *)((uint8_t *)ctx->meta_num.p + 16);
    uint64_t hist[NS_MAX_SEG + 1];
    auto seg_hist = [&](uint64_t lo) -> int {
        HIPCHK(hipMemsetAsync(d_hist, 0, (NS_MAX_SEG + 1) * 8, st));
        k_meta_hist<<<dim3((unsigned)std::min<uint64_t>(1024, (n - lo + 255) / 256 + 1)), blk, 0, st>>>(A.n_pieces, lo, n, d_hist);
        HIPCHK(hipGetLastError());
        return read_small(ctx, st, hist, d_hist, (NS_MAX_SEG + 1) * 8);
    };
    if ((rc = seg_hist(0))) return rc;
    lap("k_nseg + histogram", tt);
    tot_pieces = 0;
    uint64_t tot_seg = 0;
    for (uint32_t v = 1; v <= NS_MAX_SEG; ++v) { tot_pieces += (2ull * v - 1ull) * hist[v]; tot_seg += (uint64_t)v * hist[v]; }
    if ((rc = ensure(ctx, ctx->pieces, tot_pieces * sizeof(ns_piece) + 64)) || (rc = ensure(ctx, ctx->t_pieces, tot_pieces * sizeof(ns_piece) + 64)) ||
        (rc = ensure(ctx, ctx->t_reads, n * sizeof(ns_read))) || (rc = ensure(ctx, ctx->t_name_len, (n + 1) * 2)) ||
        (rc = ensure(ctx, ctx->t_rec_len, (n + 1) * 8)) || (rc = ensure(ctx, ctx->t_err_len, (n + 1) * 8)) ||
        (rc = ensure(ctx, ctx->accept, (n + 1) * 8)) || (rc = ensure(ctx, ctx->accept_scan, (n + 1) * 8)) ||
        (rc = ensure(ctx, ctx->key_pos, (n + 1) * 4)) || (rc = ensure(ctx, ctx->draw_x, (tot_seg + 1) * 8)) ||
        (rc = ensure(ctx, ctx->m_segptr, (n + 1) * 4)) || (rc = ensure(ctx, ctx->m_len, (tot_seg + 1) * 4)) ||
        (rc = ensure(ctx, ctx->m_species, (tot_seg + 1) * 2)) || (rc = ensure(ctx, ctx->species_bases, (size_t)ns * 8 * NS_STATS_WAYS)))
        return rc;
    HIPCHK(hipMemsetAsync(ctx->species_bases.p, 0, (size_t)ns * 8 * NS_STATS_WAYS, st));
    lap("nseg loop + buffers", tt);
    // final arrays (what the record kernels read) and the per-pass views of the same kernels
    A.f_reads = (ns_read *)ctx->reads.p; A.f_pieces = (ns_piece *)ctx->pieces.p; A.f_name_len = (uint16_t *)ctx->name_len.p;
    A.f_rec_len = (uint64_t *)ctx->rec_len.p; A.f_err_len = (uint64_t *)ctx->err_len.p;
    A.key_pos_w = (uint32_t *)ctx->key_pos.p;
    A.species_bases = (unsigned long long *)ctx->species_bases.p;
    GenArgs P = A;
    P.reads = (ns_read *)ctx->t_reads.p; P.pieces = (ns_piece *)ctx->t_pieces.p; P.name_len = (uint16_t *)ctx->t_name_len.p;
    P.rec_len = (uint64_t *)ctx->t_rec_len.p; P.err_len = (uint64_t *)ctx->t_err_len.p;
    P.accept = (uint64_t *)ctx->accept.p; P.accept_scan = (uint64_t *)ctx->accept_scan.p;
    P.draw_x = (double *)ctx->draw_x.p;
    P.m_segptr = (const uint32_t *)ctx->m_segptr.p; P.m_len = (const int32_t *)ctx->m_len.p; P.m_species = (const uint16_t *)ctx->m_species.p;
    P.list = nullptr;
    P.cap_rate = ctx->cap_rate;
    const bool lds = ctx->lds_tables && prm->kind != NS_KIND_PERFECT;
    const bool perfect = prm->kind == NS_KIND_PERFECT;      // S:838-842, 879-910: no errors, no head/tail, the quotas are never updated
    const ns_key bkey{(uint32_t)prm->seed, (uint32_t)(prm->seed >> 32), (uint32_t)prm->first_read, (uint32_t)(prm->first_read >> 32)};
    std::vector<double> cur_bases(ns, 0.0);
    std::vector<unsigned long long> sb((size_t)ns * NS_STATS_WAYS);
    uint64_t passed = 0, pieces_passed = 0, ev_base = 0;
    double ms_chain = 0;
    unsigned long long good_stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};       // counters after the last complete pass
    bool first_pass = true;
    for (uint32_t p = 0; passed < n; ++p) {
        if (p >= NS_MAX_ATTEMPT)
            return fail(ctx, NS_EINVAL, "some reads found no acceptable length within the attempt limit "
                                        "(min_len/max_len too narrow for this model, or its reads do not fit the event record: runs <= 4095 bases, "
                                            "insertion / deletion balance within +-131071 bases per segment)");
        const uint64_t m = n - passed;
        if (p && (rc = seg_hist(passed))) return rc;                               // num_segment[passed:], S:1034 — as a histogram: the reads
                                                                                   // are taken by descending segment count (S:760)
        uint64_t D = 0;
        for (uint32_t v = 1; v <= NS_MAX_SEG; ++v) D += (uint64_t)v * hist[v];
        lap("histogram", tt);
        P.attempt = p; P.draw_n = D;
        k_meta_draw<<<dim3((unsigned)((D + 255) / 256)), blk, 0, st>>>(P);         // S:852
        HIPCHK(hipGetLastError());
        lap("setup/alloc", tt);
        if ((rc = ensure_pin(ctx, ctx->pin_a, (D + 1) * 8)) || (rc = ensure_pin(ctx, ctx->pin_b, (D + 1) * 8)) ||
            (rc = ensure_pin(ctx, ctx->pin_c, (D + 1) * 8)) || (rc = ensure(ctx, ctx->draw_sel, (D + 1) * 8)) ||
            (rc = ensure(ctx, ctx->draw_sorted, (D + 1) * 8)) || (rc = ensure(ctx, ctx->meta_words, (D + 1) * 8)) ||
            false)
            return rc;
        double *h_draw = (double *)ctx->pin_a.p;
        // the filter (S:857; --perfect: S:841) on the device, order kept; sum(length_list) (S:767) is taken left to right, as Python
        // does, by the host — over the filtered values, while the device sorts them
        const MetaLenFilter flt{perfect ? (double)prm->min_len : 0.0, (double)prm->max_len, perfect};
        double *d_sel = (double *)ctx->draw_sel.p, *d_sorted = (double *)ctx->draw_sorted.p;
        {
            size_t tmp = 0;
            HIPCHK(hipcub::DeviceSelect::If(nullptr, tmp, P.draw_x, d_sel, (int *)ctx->meta_num.p, (int)D, flt, st));
            if ((rc = ensure(ctx, ctx->scan_tmp, tmp))) return rc;
            HIPCHK(hipcub::DeviceSelect::If(ctx->scan_tmp.p, tmp, P.draw_x, d_sel, (int *)ctx->meta_num.p, (int)D, flt, st));
        }
        int v_sel = 0;
        if ((rc = read_small(ctx, st, &v_sel, ctx->meta_num.p, 4))) return rc;
        lap("draw + filter", tt);
        const uint64_t V = (uint64_t)v_sel;
        if (!V) continue;                                                          // S:858-859
        uint64_t chim = 0;
        for (uint32_t v = 2; v <= NS_MAX_SEG; ++v) chim += (uint64_t)v * hist[v];  // S:761: the first `chim` lengths keep their order
        if (chim > V) chim = V;
        HIPCHK(hipMemcpyAsync(h_draw, d_sel, V * 8, hipMemcpyDeviceToHost, st));
        hipEvent_t ev_draws = ctx->evt[11];                                        // (the filtered draws have reached the host)
        HIPCHK(hipEventRecord(ev_draws, st));
        if (chim) HIPCHK(hipMemcpyAsync(d_sorted, d_sel, chim * 8, hipMemcpyDeviceToDevice, st));
        if (V > chim) {                                                            // S:764-765
            size_t tmp = 0;
            HIPCHK(hipcub::DeviceRadixSort::SortKeysDescending(nullptr, tmp, d_sel + chim, d_sorted + chim, (int)(V - chim), 0, 64, st));
            if ((rc = ensure(ctx, ctx->scan_tmp, tmp))) return rc;
            HIPCHK(hipcub::DeviceRadixSort::SortKeysDescending(ctx->scan_tmp.p, tmp, d_sel + chim, d_sorted + chim, (int)(V - chim), 0, 64, st));
        }
        k_meta_words<<<dim3((unsigned)((V + 255) / 256)), blk, 0, st>>>(P, (uint2 *)ctx->meta_words.p, V);
        HIPCHK(hipGetLastError());
        double *h_sorted = (double *)ctx->pin_b.p;
        uint2 *h_words = (uint2 *)ctx->pin_c.p;
        HIPCHK(hipMemcpyAsync(h_sorted, d_sorted, V * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(h_words, ctx->meta_words.p, V * 8, hipMemcpyDeviceToHost, st));
        P.m_reversed = u32_to_p(ns_draw(bkey, ST_STRAND, 0, p, 0, 0).x) > ctx->m.strandness_rate ? 1u : 0u;   // S:860
        // S:862-865: the reads that get their lengths are the first np of the descending segment-count order; their first segment / first
        // piece are closed forms of the histogram (k_meta_layout).  The error lists of a read do not depend on its species, so the pass
        // launches them NOW, for the reads the V lengths cover (np_spec: all that assign_species can reach, S:781-782), and the host walks the
        // quotas (sum(length_list) + assign_species: ~3 ms per 10^6 reads, sequential by definition) while they run; k_meta_tail then takes
        // the reads the walk did assign (np <= np_spec: fewer only when every quota is used up) through positions and acceptance.
        uint64_t np_spec = 0;
        { uint64_t ptr = 0; bool stop = false;
          for (int v = (int)NS_MAX_SEG; v >= 1 && !stop; --v) {
              const uint64_t fit = (V - ptr) / (uint64_t)v, c = std::min<uint64_t>(hist[v], fit);
              np_spec += c; ptr += c * (uint64_t)v;
              if (c < hist[v]) stop = true;
          } }
        np_spec = std::min<uint64_t>(np_spec, m);
        auto layout = [&](uint64_t np_l, MetaHist &H, uint64_t &sp, uint64_t &po) {
            uint64_t left = np_l; sp = 0; po = 0;
            H.cnt[0] = 0;
            for (int v = (int)NS_MAX_SEG; v >= 1; --v) {
                const uint64_t c = std::min<uint64_t>(hist[v], left);
                H.cnt[v] = (uint32_t)c; left -= c; sp += c * (uint64_t)v; po += c * (2ull * (uint64_t)v - 1ull);
            }
        };
        uint64_t pass_cap = 0, np64 = 0, sp = 0, po = 0;
        double to_add = 0;
        size_t np = 0;
        bool walked = false;
        uint16_t *h_species = (uint16_t *)h_draw;                                  // (once the sum is taken the draws are no longer needed: reuse the staging)
        for (int retry = 0;; ++retry) {
            const size_t np_l = walked ? np : (size_t)np_spec;                     // the reads of this launch
            if (!np_l) break;
            MetaHist H;
            uint64_t sp_l, po_l;
            layout(np_l, H, sp_l, po_l);
            k_meta_layout<<<dim3((unsigned)((np_l + 1 + 255) / 256)), blk, 0, st>>>(H, (uint32_t)np_l, (uint32_t *)ctx->m_segptr.p, (uint32_t *)ctx->piece_off.p);
            HIPCHK(hipGetLastError());
            k_meta_round<<<dim3((unsigned)((sp_l + 255) / 256)), blk, 0, st>>>(d_sorted, (int32_t *)ctx->m_len.p, sp_l);   // S:871: int(round(length))
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemsetAsync(P.accept, 0, (np_l + 1) * 8, st));
            HIPCHK(hipMemsetAsync(P.ev_cap + np_l, 0, 8, st));
            P.list_n = (uint32_t)np_l;
            P.defer_tail = 1;
            const dim3 grid_l((unsigned)((np_l + 255) / 256));
            k_lengths<false><<<grid_l, blk, 0, st>>>(P);                           // gaps, head/tail, planned pieces (S:872, 898-903)
            HIPCHK(hipGetLastError());
            if ((rc = scan_u64(ctx, P.ev_cap, P.ev_off, np_l + 1))) return rc;
            if (!walked) {               // S:767: sum(length_list), left to right as Python adds (~1 ms per 10^6 values) — while the device sorts and plans
                HIPCHK(hipEventSynchronize(ev_draws));
                for (uint64_t j = 0; j < V; ++j) to_add += h_draw[j];
                lap("sum(length_list)", tt);
            }
            if ((rc = read_small(ctx, st, &pass_cap, P.ev_off + np_l, 8))) return rc;   // also: the sorted lengths and the words have reached the host
            if ((rc = ensure_keep(ctx, ctx->events, (size_t)(ev_base + pass_cap) * sizeof(ns_event) + 64, (size_t)ev_base * sizeof(ns_event)))) return rc;
            P.events = (ns_event *)ctx->events.p; P.ev_base = ev_base;
            P.m_passed = (uint32_t)passed; P.m_pieces_passed = (uint32_t)pieces_passed;
            HIPCHK(hipEventRecord(ctx->evt[3], st));
            if (first_pass && retry == 0) { HIPCHK(hipEventRecord(ctx->evt[2], st)); first_pass = false; }
            // the pass positions are (nearly) sorted by descending length: the head of the list goes to the cooperative chain
            uint32_t n_coop = 0;
            if (ctx->coop_ok && !perfect && np_l >= ctx->coop_min) n_coop = (uint32_t)(np_l >> ctx->coop_shift);
            GenArgs Q = P;
            if (n_coop) {
                GenArgs B = P; B.list_n = n_coop; B.list_base = 0;
                HIPCHK(hipEventRecord(ctx->ev_fork, st));
                HIPCHK(hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
                B.coop_mix = (size_t)B.m.ct.n_words_mix * 8 <= 32u * 1024u ? 1u : 0u;
                k_chain<false, true><<<dim3(n_coop), dim3(64), B.coop_mix ? (size_t)B.m.ct.n_words_mix * 8 : 0, ctx->stream2>>>(B);
                HIPCHK(hipGetLastError());
                HIPCHK(hipEventRecord(ctx->ev_join, ctx->stream2));
                Q.list_base = n_coop; Q.list_n = (uint32_t)np_l - n_coop;
            }
            const uint32_t cb = lds ? ctx->chain_block : NS_CHAIN_BLOCK;
            const dim3 grid_pc((unsigned)((Q.list_n + cb - 1) / cb)), blk_c(cb);
            if (lds) k_chain<true, false><<<grid_pc, blk_c, ctx->lds_bytes + (Q.ev_stage ? cb * 32u : 0u), st>>>(Q);
            else k_chain<false, false><<<grid_pc, blk_c, 0, st>>>(Q);
            HIPCHK(hipGetLastError());
            if (n_coop) HIPCHK(hipStreamWaitEvent(st, ctx->ev_join, 0));
            HIPCHK(hipEventRecord(ctx->evt[4], st));
            if (!walked) {               // ---- the host's walk over the quotas, next to the chain kernels
                const uint64_t P_seg = assign_species_host(ctx, h_sorted, V, to_add, h_words, hist, cur_bases, h_species, &np64);   // S:866-867
                (void)P_seg;
                lap("assign_species", tt);
                np = (size_t)std::min<uint64_t>(np64, m);
                walked = true;
                if (np > np_l) return fail(ctx, NS_ESTATE, "metagenome pass: assign_species reached more reads than the pass planned");
            }
            if (!np) {                   // every quota is used up: nothing of this launch counts
                fold_stats(ctx, st);
                HIPCHK(hipMemcpyAsync(ctx->stats.p, good_stats, 8 * sizeof(unsigned long long), hipMemcpyHostToDevice, st));
                break;
            }
            MetaHist Hn;
            layout(np, Hn, sp, po);                          // (the first np reads of the launch: the layout of a prefix is a prefix of the layout)
            HIPCHK(hipMemcpyAsync(ctx->m_species.p, h_species, sp * 2, hipMemcpyHostToDevice, st));
            HIPCHK(hipMemsetAsync(P.accept + np, 0, 8, st)); // (a read beyond np that the launch left pending is not a read of this pass)
            P.list_n = (uint32_t)np;
            k_meta_tail<<<dim3((unsigned)((np + 255) / 256)), blk, 0, st>>>(P);
            HIPCHK(hipGetLastError());
            fold_stats(ctx, st);
            if ((rc = read_small(ctx, st, stats, ctx->stats.p, 8 * sizeof(unsigned long long)))) return rc;
            HIPCHK(hipEventElapsedTime(&ms, ctx->evt[3], ctx->evt[4])); ms_chain += ms;
            if (!(stats[0] & NS_OVER_MASK)) break;
            // a read outgrew its event capacity (rare): the lists of the pass are repeated with twice the capacity
            info->n_overflow += stats[0] & NS_OVER_MASK;
            if (retry >= 6) return fail(ctx, NS_ENOMEM, "event capacity overflow persists after 6 retries");
            P.cap_rate *= 2.0; P.cap_gap_mul *= 2;
            HIPCHK(hipMemcpyAsync(ctx->stats.p, good_stats, 8 * sizeof(unsigned long long), hipMemcpyHostToDevice, st));
        }
        if (!np) continue;
        const dim3 grid_p((unsigned)((np + 255) / 256));
        if (P.hp) {            // -k: the homopolymer stage decides the final length, and with it whether the pass accepts the read (S:1023-1024)
            GenArgs H = P;
            H.prm.n_reads = np;                      // bound of the thread-per-read kernels of the stage
            double ms_hp = 0;
            if ((rc = hp_stage1(ctx, prm, H, np, tot_pieces, ev_base + pass_cap, stats, &ms_hp))) return rc;
            info->ms_kernel[NS_K_HP] += ms_hp;
            stats[5] = 0;
        } else if (prm->emit_errlog) {     // sizes of the error-profile rows of the reads this pass accepted (k_meta_commit adds the read numbers)
            GenArgs H = P;
            H.prm.n_reads = np;
            k_errlen<<<dim3((unsigned)((np + NS_WPB - 1) / NS_WPB)), dim3(64 * NS_WPB), 0, st>>>(H);
            HIPCHK(hipGetLastError());
        }
        memcpy(good_stats, stats, sizeof good_stats);
        // the pieces of the reads this pass accepts go behind those of the earlier passes: a pass re-sorts the remaining segment counts, so
        // the total over all passes can exceed the first plan (sum over the reads in their original order)
        if ((rc = ensure_keep(ctx, ctx->pieces, (size_t)(pieces_passed + po) * sizeof(ns_piece) + 64, (size_t)pieces_passed * sizeof(ns_piece)))) return rc;
        A.f_pieces = P.f_pieces = (ns_piece *)ctx->pieces.p;
        if ((rc = scan_u64(ctx, P.accept, P.accept_scan, np + 1))) return rc;
        k_meta_commit<<<grid_p, blk, 0, st>>>(P);
        HIPCHK(hipGetLastError());
        uint64_t acc = 0;
        HIPCHK(hipMemcpyAsync(&acc, P.accept_scan + np, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(sb.data(), ctx->species_bases.p, (size_t)ns * 8 * NS_STATS_WAYS, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        lap("plan/chain/commit", tt);
        for (uint32_t s = 0; s < ns && !perfect; ++s) {
            unsigned long long tot = 0;
            for (uint32_t w = 0; w < NS_STATS_WAYS; ++w) tot += sb[(size_t)w * ns + s];
            cur_bases[s] = (double)tot;
        }
        passed += acc & 0xffffffffull; pieces_passed += acc >> 32;
        ev_base += pass_cap;
    }
    if (first_pass) HIPCHK(hipEventRecord(ctx->evt[2], st));
    A.events = (ns_event *)ctx->events.p;
    A.pieces = (ns_piece *)ctx->pieces.p;
    A.key_pos = (const uint32_t *)ctx->key_pos.p;
    A.m_species = P.m_species; A.m_segptr = P.m_segptr;
    tot_pieces = pieces_passed;
    tot_cap = ev_base;
    ctx->last_species_bases = cur_bases;
    info->ms_kernel[NS_K_EVENTS] = ms_chain;
    return NS_OK;
}

// The aligned / --perfect reads of a transcriptome batch (S:1080-1104; kernels: k_trx_picks ... k_trx_commit above): the blocks of
// NS_TRX_BLOCK read indices that overlap the batch are walked from their starts, every candidate is tried once (k_lengths + k_chain over
// the candidate table), and the survivors that belong to the batch move to their slots.  Too few picks for a full table, or too few
// survivors in a block, repeats the planning with more picks / a longer table (deterministic: a pick is keyed by (block, pick), a
// candidate by (block, candidate)).
static int trx_passes(ns_ctx *ctx, const ns_params *prm, ns_batch_info *info, GenArgs &A, uint64_t &tot_pieces, uint64_t &tot_cap,
                      unsigned long long *stats) {
    const size_t n = (size_t)prm->n_reads;
    hipStream_t st = ctx->stream;
    const dim3 blk(256);
    const uint64_t W = NS_TRX_BLOCK, g0 = prm->first_read, b0 = g0 / W, nb = (g0 + n - 1) / W - b0 + 1;
    int rc;
    float ms = 0;
    if (ctx->tx.n_expr >= (1u << 22)) return fail(ctx, NS_EINVAL, "transcriptome: more than 2^22 expressed transcripts");
    HIPCHK(hipEventRecord(ctx->evt[1], st));
    A.key_first = b0 * W;                                   // the keys of the candidate table count from the first block
    double ms_chain = 0;
    bool planned = false;
    for (int round = 0;; ++round) {
        if (round >= 24) return fail(ctx, NS_EINVAL, "transcriptome: no aligned length below the transcript length within the pick limit "
                                                     "(2-D KDE and transcript lengths do not match)");
        const uint32_t C = (uint32_t)std::min<uint64_t>(2 * W, W + ctx->trx_margin);
        uint64_t M64 = ((uint64_t)C * ctx->trx_pick_pct / 100 + 64 + 63) & ~63ull;
        if (M64 >= (1ull << NS_TRX_PICK_BITS)) M64 = (1ull << NS_TRX_PICK_BITS) - 64;
        const uint32_t M = (uint32_t)M64;
        const uint64_t n_picks = nb * M, np = nb * C;
        if (np > 0x7ffffff0ull || n_picks > 0x7ffffff0ull) return fail(ctx, NS_EINVAL, "transcriptome batch too large (split into several calls)");
        if ((rc = ensure(ctx, ctx->trx_pick_e, n_picks * 4 + 64)) || (rc = ensure(ctx, ctx->trx_pick_y, n_picks * 4 + 64)) ||
            (rc = ensure(ctx, ctx->trx_keys, n_picks * 8 + 64)) || (rc = ensure(ctx, ctx->trx_keys2, n_picks * 8 + 64)) ||
            (rc = ensure(ctx, ctx->trx_prev, n_picks * 4 + 64)) || (rc = ensure(ctx, ctx->trx_cand, np * 4 + 64)) ||
            (rc = ensure(ctx, ctx->meta_num, 64)))
            return rc;
        unsigned long long *d_short = (unsigned long long *)ctx->meta_num.p;
        HIPCHK(hipMemsetAsync(d_short, 0, 16, st));
        k_trx_picks<<<dim3((unsigned)((n_picks + 255) / 256)), blk, 0, st>>>(A, n_picks, M, b0, (uint32_t *)ctx->trx_pick_e.p, (int32_t *)ctx->trx_pick_y.p,
                                                                              (uint64_t *)ctx->trx_keys.p);
        HIPCHK(hipGetLastError());
        {
            int end_bit = 22 + (int)NS_TRX_PICK_BITS;
            for (uint64_t v = nb - 1; v; v >>= 1) ++end_bit;
            size_t tmp = 0;
            HIPCHK(hipcub::DeviceRadixSort::SortKeys(nullptr, tmp, (const uint64_t *)ctx->trx_keys.p, (uint64_t *)ctx->trx_keys2.p, (int)n_picks, 0, end_bit, st));
            if ((rc = ensure(ctx, ctx->scan_tmp, tmp))) return rc;
            HIPCHK(hipcub::DeviceRadixSort::SortKeys(ctx->scan_tmp.p, tmp, (const uint64_t *)ctx->trx_keys.p, (uint64_t *)ctx->trx_keys2.p, (int)n_picks, 0, end_bit, st));
        }
        k_trx_prev<<<dim3((unsigned)((n_picks + 255) / 256)), blk, 0, st>>>((const uint64_t *)ctx->trx_keys2.p, n_picks, M, (int32_t *)ctx->trx_prev.p);
        k_trx_walk<<<dim3((unsigned)nb), dim3(64), M, st>>>(M, C, (const int32_t *)ctx->trx_prev.p, (const int32_t *)ctx->trx_pick_y.p,
                                                            (uint32_t *)ctx->trx_cand.p, d_short);
        HIPCHK(hipGetLastError());
        unsigned long long n_short = 0;
        if ((rc = read_small(ctx, st, &n_short, d_short, 8))) return rc;
        if (n_short) {                                       // picks ran out before the table of some block was full: more picks per candidate
            if (M64 + 64 >= (1ull << NS_TRX_PICK_BITS)) return fail(ctx, NS_EINVAL, "transcriptome: pick limit reached (almost no pick gives an aligned length below its transcript)");
            ctx->trx_pick_pct *= 2;
            continue;
        }
        // ---- the candidate table: one try per position
        if ((rc = ensure(ctx, ctx->pieces, np * sizeof(ns_piece) + 64)) || (rc = ensure(ctx, ctx->t_pieces, np * sizeof(ns_piece) + 64)) ||
            (rc = ensure(ctx, ctx->t_reads, np * sizeof(ns_read))) || (rc = ensure(ctx, ctx->t_name_len, (np + 1) * 2)) ||
            (rc = ensure(ctx, ctx->t_rec_len, (np + 1) * 8)) || (rc = ensure(ctx, ctx->t_err_len, (np + 1) * 8)) ||
            (rc = ensure(ctx, ctx->accept, (np + 1) * 8)) || (rc = ensure(ctx, ctx->accept_scan, (np + 1) * 8)) ||
            (rc = ensure(ctx, ctx->key_pos, (n + 1) * 4)) || (rc = ensure(ctx, ctx->t_polya, (np + 1) * 2)) ||
            (rc = ensure(ctx, ctx->ev_cap, (np + 1) * 8)) || (rc = ensure(ctx, ctx->ev_off, (np + 1) * 8)) ||
            (rc = ensure(ctx, ctx->sort_key, (np + 1) * 4)) || (rc = ensure(ctx, ctx->sort_idx, (np + 1) * 4)) ||
            (rc = ensure(ctx, ctx->sort_key_out, (np + 1) * 4)) || (rc = ensure(ctx, ctx->list_b, (np + 1) * 4)) ||
            (A.ir_need && (rc = ensure(ctx, ctx->t_ir_need, (np + 1) * 8))))
            return rc;
        A.ev_cap = (uint64_t *)ctx->ev_cap.p; A.ev_off = (uint64_t *)ctx->ev_off.p;
        A.sort_key = (uint32_t *)ctx->sort_key.p; A.sort_idx = (uint32_t *)ctx->sort_idx.p;
        A.f_reads = (ns_read *)ctx->reads.p; A.f_pieces = (ns_piece *)ctx->pieces.p; A.f_name_len = (uint16_t *)ctx->name_len.p;
        A.f_rec_len = (uint64_t *)ctx->rec_len.p; A.f_err_len = (uint64_t *)ctx->err_len.p;
        A.key_pos_w = (uint32_t *)ctx->key_pos.p;
        GenArgs P = A;
        P.reads = (ns_read *)ctx->t_reads.p; P.pieces = (ns_piece *)ctx->t_pieces.p; P.name_len = (uint16_t *)ctx->t_name_len.p;
        P.rec_len = (uint64_t *)ctx->t_rec_len.p; P.err_len = (uint64_t *)ctx->t_err_len.p;
        P.accept = (uint64_t *)ctx->accept.p; P.accept_scan = (uint64_t *)ctx->accept_scan.p;
        P.polya = (uint16_t *)ctx->t_polya.p;
        if (A.ir_need) P.ir_need = (uint64_t *)ctx->t_ir_need.p;
        P.trx_C = C; P.trx_M = M; P.trx_cand = (const uint32_t *)ctx->trx_cand.p;
        P.trx_pick_e = (const uint32_t *)ctx->trx_pick_e.p; P.trx_pick_y = (const int32_t *)ctx->trx_pick_y.p;
        P.list = nullptr; P.list_n = (uint32_t)np; P.list_base = 0; P.attempt = 0; P.l_off = nullptr; P.ev_base = 0;
        P.cap_rate = ctx->cap_rate;
        const bool lds = ctx->lds_tables && prm->kind != NS_KIND_PERFECT;
        const dim3 grid_p((unsigned)((np + 255) / 256));
        uint64_t cap = 0;
        for (int retry = 0;; ++retry) {
            HIPCHK(hipMemsetAsync(ctx->stats.p, 0, NS_STATS_BYTES, st));
            HIPCHK(hipMemsetAsync(P.accept, 0, (np + 1) * 8, st));
            HIPCHK(hipMemsetAsync(P.polya, 0, (np + 1) * 2, st));
            if (P.ir_need) HIPCHK(hipMemsetAsync(P.ir_need, 0, (np + 1) * 8, st));
            HIPCHK(hipMemsetAsync(P.ev_cap + np, 0, 8, st));
            P.list = nullptr;
            k_lengths<false><<<grid_p, blk, 0, st>>>(P);
            HIPCHK(hipGetLastError());
            if ((rc = scan_u64(ctx, P.ev_cap, P.ev_off, np + 1))) return rc;
            if ((rc = visiting_order(ctx, P.sort_key, P.sort_idx, np, (uint32_t *)ctx->list_b.p))) return rc;      // candidates by descending length
            if (!planned) { HIPCHK(hipEventRecord(ctx->evt[2], st)); planned = true; }
            if ((rc = read_small(ctx, st, &cap, P.ev_off + np, 8))) return rc;
            if ((rc = ensure(ctx, ctx->events, (size_t)cap * sizeof(ns_event) + 64))) return rc;
            A.events = P.events = (ns_event *)ctx->events.p;
            P.list = (const uint32_t *)ctx->list_b.p;
            HIPCHK(hipEventRecord(ctx->evt[3], st));
            const uint32_t cb = lds ? ctx->chain_block : NS_CHAIN_BLOCK;
            const dim3 grid_c((unsigned)((np + cb - 1) / cb)), blk_c(cb);
            if (lds) k_chain<true, false><<<grid_c, blk_c, ctx->lds_bytes + (P.ev_stage ? cb * 32u : 0u), st>>>(P);
            else k_chain<false, false><<<grid_c, blk_c, 0, st>>>(P);
            HIPCHK(hipGetLastError());
            HIPCHK(hipEventRecord(ctx->evt[4], st));
            fold_stats(ctx, st);
            if ((rc = read_small(ctx, st, stats, ctx->stats.p, 8 * sizeof(unsigned long long)))) return rc;
            HIPCHK(hipEventElapsedTime(&ms, ctx->evt[3], ctx->evt[4])); ms_chain += ms;
            if (!(stats[0] & NS_OVER_MASK)) break;
            info->n_overflow += stats[0] & NS_OVER_MASK;         // a read outgrew its event capacity (rare): again with twice the capacity
            if (retry >= 6) return fail(ctx, NS_ENOMEM, "event capacity overflow persists after 6 retries");
            P.cap_rate *= 2.0; P.cap_gap_mul *= 2;
        }
        if ((rc = scan_u64(ctx, P.accept, P.accept_scan, np + 1))) return rc;
        if (A.ir_need) HIPCHK(hipMemsetAsync(A.ir_need, 0, (n + 1) * 8, st));
        HIPCHK(hipMemsetAsync(A.rec_len + n, 0, 8, st));     // the sentinels of the scans over the final reads
        HIPCHK(hipMemsetAsync(A.err_len + n, 0, 8, st));
        HIPCHK(hipMemsetAsync(d_short, 0, 16, st));
        HIPCHK(hipMemsetAsync((unsigned long long *)ctx->stats.p + 1, 0, 3 * sizeof(unsigned long long), st));
        k_trx_commit<<<grid_p, blk, 0, st>>>(P, np, b0, g0, (uint64_t)n, (uint16_t *)ctx->polya.p, A.ir_need, d_short);
        HIPCHK(hipGetLastError());
        fold_stats(ctx, st);
        if ((rc = read_small(ctx, st, &n_short, d_short, 8, stats, ctx->stats.p, 8 * sizeof(unsigned long long)))) return rc;
        if (n_short) {                                       // a block lost more candidates than the table has to spare: a longer table
            if (C >= 2 * W) return fail(ctx, NS_EINVAL, "transcriptome: more than half of the candidates of a block overshoot their transcript");
            ctx->trx_margin = (uint32_t)std::min<uint64_t>(W, 2ull * ctx->trx_margin);
            continue;
        }
        tot_cap = cap;
        break;
    }
    A.pieces = (ns_piece *)ctx->pieces.p;
    A.key_pos = (const uint32_t *)ctx->key_pos.p;
    tot_pieces = n;
    info->ms_kernel[NS_K_EVENTS] = ms_chain;
    return NS_OK;
}

static void lend_tables(const ns_ctx *ctx, ns_ctx *c);
int ns_generate(ns_ctx *ctx, const ns_params *prm, ns_batch_info *info) {
    if (!ctx) return NS_EINVAL;
    if (!prm || !info) return fail(ctx, NS_EINVAL, "null params/info");
    // a step companion called directly (include/nanosim_amd.h allows it): the owner may have loaded another model or reference since the
    // tables were lent — the pointers the companion holds by value would be freed memory
    if (ctx->borrowed && ctx->owner) lend_tables(ctx->owner, ctx);
    if (!ctx->has_model || !ctx->has_ref) return fail(ctx, NS_ESTATE, "ns_generate before ns_load_model/ns_set_reference");
    if (prm->kind > NS_KIND_PERFECT) return fail(ctx, NS_EINVAL, "bad kind");
    if (prm->emit_records > NS_EMIT_SIZES) return fail(ctx, NS_EINVAL, "bad emit_records");
    if (prm->kind != NS_KIND_PERFECT && !(ctx->m.flags & NS_MODEL_HAS_ERRORS)) return fail(ctx, NS_EINVAL, "model has no error tables");
    if (prm->kind == NS_KIND_UNALIGNED && !prm->use_lognormal && !(ctx->m.flags & NS_MODEL_HAS_UNALIGNED))
        return fail(ctx, NS_EINVAL, "model has no unaligned-length KDE");
    if (prm->fastq && !(ctx->m.flags & NS_MODEL_HAS_QUALS)) return fail(ctx, NS_EINVAL, "model has no quality tables");
    if (prm->chimeric && !(ctx->m.flags & NS_MODEL_HAS_CHIMERIC)) return fail(ctx, NS_EINVAL, "model has no chimeric tables");
    const bool hp_on = prm->kmer_bias && prm->kind == NS_KIND_ALIGNED;      // S:1413: only aligned segments; --perfect never
    if (hp_on && !(ctx->m.flags & NS_MODEL_HAS_HP)) return fail(ctx, NS_EINVAL, "-k needs the homopolymer model (-hp)");
    const bool meta_al = prm->meta && prm->kind != NS_KIND_UNALIGNED;       // aligned or --perfect worker of simulation_aligned_metagenome
    if (prm->meta) {
        if (!ctx->nspecies) return fail(ctx, NS_ESTATE, "metagenome batch before ns_set_species");
        if (prm->kind == NS_KIND_PERFECT && prm->chimeric) return fail(ctx, NS_EINVAL, "perfect reads cannot be chimeric");
        if (meta_al && !ctx->has_abun) return fail(ctx, NS_ESTATE, "metagenome batch before ns_set_abundance");
        if (meta_al && prm->chimeric && !ctx->has_inflated) return fail(ctx, NS_EINVAL, "chimeric metagenome batch needs abun_inflated");
    }
    if (prm->trx) {
        if (!ctx->has_trx) return fail(ctx, NS_ESTATE, "transcriptome batch before ns_set_transcriptome");
        if (prm->meta || prm->chimeric || prm->use_lognormal) return fail(ctx, NS_EINVAL, "transcriptome batches are neither metagenome nor chimeric nor log-normal");
        if (prm->kind != NS_KIND_UNALIGNED && !(ctx->m.flags & NS_MODEL_HAS_KDE2D)) return fail(ctx, NS_EINVAL, "model has no 2-D KDE (_aligned_region_2d)");
    }
    if (prm->model_ir) {
        if (!prm->trx) return fail(ctx, NS_EINVAL, "model_ir is a transcriptome option");
        if (!ctx->has_ir) return fail(ctx, NS_ESTATE, "model_ir batch before ns_set_intron_retention");
    }
    const bool ir_on = prm->model_ir && prm->kind == NS_KIND_ALIGNED;       // S:1156: not for --perfect, not for unaligned reads
    if (prm->n_reads > 0x7ffffff0ull) return fail(ctx, NS_EINVAL, "batch too large (split into several calls)");
    if (prm->first_read + prm->n_reads >= (1ull << 40)) return fail(ctx, NS_EINVAL, "read index exceeds 2^40");
    HIPCHK(hipSetDevice(ctx->device));
    const size_t n = (size_t)prm->n_reads;
    memset(info, 0, sizeof *info);
    ctx->has_batch = false;
    if (!n) { ctx->last = *info; ctx->has_batch = true; return NS_OK; }
    int rc;
    if ((rc = ensure(ctx, ctx->n_pieces, (n + 1) * 4)) || (rc = ensure(ctx, ctx->piece_off, (n + 1) * 4)) ||
        (rc = ensure(ctx, ctx->ev_cap, (n + 1) * 8)) || (rc = ensure(ctx, ctx->ev_off, (n + 1) * 8)) || (rc = ensure(ctx, ctx->l_cap, (n + 1) * 8)) || (rc = ensure(ctx, ctx->l_off, (n + 1) * 8)) ||
        (rc = ensure(ctx, ctx->rec_len, (n + 1) * 8)) || (rc = ensure(ctx, ctx->rec_off, (n + 1) * 8)) ||
        (rc = ensure(ctx, ctx->err_len, (n + 1) * 8)) || (rc = ensure(ctx, ctx->err_off, (n + 1) * 8)) ||
        (rc = ensure(ctx, ctx->name_len, (n + 1) * 2)) || (rc = ensure(ctx, ctx->reads, n * sizeof(ns_read))) ||
        (rc = ensure(ctx, ctx->stats, NS_STATS_BYTES)) ||
        (rc = ensure(ctx, ctx->sort_key, (n + 1) * 4)) || (rc = ensure(ctx, ctx->sort_idx, (n + 1) * 4)) ||
        (rc = ensure(ctx, ctx->sort_key_out, (n + 1) * 4)) || (rc = ensure(ctx, ctx->order, (n + 1) * 4)) ||
        (rc = ensure(ctx, ctx->list_b, (n + 1) * 4)) || (rc = ensure(ctx, ctx->list_c, (n + 1) * 4)) || (rc = ensure(ctx, ctx->rstate, (n + 1) * 4)) ||
        (rc = ensure(ctx, ctx->att_base, (n + 1) * 4)) || (rc = ensure(ctx, ctx->scr_len, (n + 1) * 8)) ||
        (rc = ensure(ctx, ctx->scr_off, (n + 1) * 8)))
        return rc;

    GenArgs A;
    memset(&A, 0, sizeof A);
    A.prm = *prm; A.m = ctx->m; A.ref = ctx->ref;
    A.key_first = A.name_first = prm->first_read;
    A.cap_gap_mul = 2;
    // events of the thread-per-read chain staged four at a time in LDS, when the tables leave room for it next to four workgroups per CU
    A.ev_stage = (ctx->lds_tables && ctx->lds_bytes + ctx->chain_block * 32u <= 64u * 1024u && !getenv("NS_NO_EV_STAGE")) ? (uint32_t)ctx->lds_bytes : 0u;
    A.n_pieces = (uint32_t *)ctx->n_pieces.p; A.piece_off = (uint32_t *)ctx->piece_off.p;
    A.ev_cap = (uint64_t *)ctx->ev_cap.p; A.ev_off = (uint64_t *)ctx->ev_off.p; A.l_cap = (uint64_t *)ctx->l_cap.p;
    A.rec_len = (uint64_t *)ctx->rec_len.p; A.rec_off = (uint64_t *)ctx->rec_off.p;
    A.err_len = (uint64_t *)ctx->err_len.p; A.err_off = (uint64_t *)ctx->err_off.p;
    A.name_len = (uint16_t *)ctx->name_len.p; A.reads = (ns_read *)ctx->reads.p;
    A.stats = (unsigned long long *)ctx->stats.p;
    A.sort_key = (uint32_t *)ctx->sort_key.p; A.sort_idx = (uint32_t *)ctx->sort_idx.p;
    A.rstate = (uint32_t *)ctx->rstate.p; A.att_base = (uint32_t *)ctx->att_base.p;
    A.scr_len = (uint64_t *)ctx->scr_len.p; A.scr_off = (uint64_t *)ctx->scr_off.p;
    A.hp = hp_on ? 1u : 0u; A.keep_state = 0;
    A.errlen_later = prm->emit_errlog ? 1u : 0u;
    A.dbg = ctx->dbg;
    if (prm->trx) {
        if ((rc = ensure(ctx, ctx->polya, (n + 1) * 2))) return rc;
        A.tx = ctx->tx; A.polya = (uint16_t *)ctx->polya.p;
        HIPCHK(hipMemsetAsync(ctx->polya.p, 0, (n + 1) * 2, ctx->stream));
    }
    ctx->spliced_bytes = 0;
    if (ir_on) {
        if ((rc = ensure(ctx, ctx->ir_need, (n + 1) * 8)) || (rc = ensure(ctx, ctx->ir_off, (n + 1) * 8))) return rc;
        A.ir = ctx->ir; A.ir_need = (uint64_t *)ctx->ir_need.p;
    }
    A.meta = prm->meta ? 1u : 0u; A.nspecies = ctx->nspecies; A.species_chrom_off = (const uint32_t *)ctx->species_chrom_off.p;
    A.next_n = (uint32_t *)((unsigned long long *)ctx->stats.p + 6);
    uint32_t *list_a = (uint32_t *)ctx->order.p, *list_b = (uint32_t *)ctx->list_b.p, *list_c = (uint32_t *)ctx->list_c.p;
    const dim3 blk(256);
    const dim3 grid_t((unsigned)((n + 1 + 255) / 256));        // thread-per-read kernels (n+1 for the scan sentinel)
    const dim3 grid_w((unsigned)((n + NS_WPB - 1) / NS_WPB)), blk_w(64 * NS_WPB);     // wave-per-read kernels
    hipStream_t st = ctx->stream;
    unsigned long long stats[8];
    uint64_t tot_pieces = 0, tot_cap = 0;
    double cap_rate = ctx->cap_rate;
    const bool lds = ctx->lds_tables && prm->kind != NS_KIND_PERFECT;
    float ms = 0;
    ctx->rec_timed = false;
    HIPCHK(hipEventRecord(ctx->evt[0], st));
    double ms_hp = 0;
    const bool trx_tab = prm->trx && prm->kind != NS_KIND_UNALIGNED;        // transcript + aligned length per block walk (trx_passes)
    if (meta_al && (rc = meta_passes(ctx, prm, info, A, tot_pieces, tot_cap, stats))) return rc;
    if (meta_al && A.hp) {         // the passes validated the final lengths; the stage runs once more on the reads in their final order
        ms_hp = info->ms_kernel[NS_K_HP];
        if ((rc = hp_stage1(ctx, prm, A, n, tot_pieces, tot_cap, stats, &ms_hp))) return rc;
    }
    auto ir_splice = [&]() -> int {          // splice arena: slot offsets, then the copy from the genome (before anything reads the pieces' bases)
        if ((rc = scan_u64(ctx, A.ir_need, (uint64_t *)ctx->ir_off.p, n + 1))) return rc;
        uint64_t arena_bytes = 0;
        if ((rc = read_small(ctx, st, &arena_bytes, (uint64_t *)ctx->ir_off.p + n, 8))) return rc;
        if ((rc = ensure(ctx, ctx->spliced, (size_t)arena_bytes + 64))) return rc;
        A.ir.arena = (uint8_t *)ctx->spliced.p; A.ir.arena_off = (const uint64_t *)ctx->ir_off.p;
        A.ref.spliced = (const uint8_t *)ctx->spliced.p;
        ctx->spliced_bytes = arena_bytes;
        if (arena_bytes) {
            k_ir_splice<<<grid_w, blk_w, 0, st>>>(A);
            HIPCHK(hipGetLastError());
        }
        return NS_OK;
    };
    if (trx_tab) {
        if ((rc = trx_passes(ctx, prm, info, A, tot_pieces, tot_cap, stats))) return rc;
        if (ir_on && (rc = ir_splice())) return rc;
        if (A.hp && (rc = hp_stage1(ctx, prm, A, n, tot_pieces, tot_cap, stats, &ms_hp))) return rc;    // (no length limits on these reads: nothing fails S:1429)
    }
    for (int hp_round = 0; !meta_al && !trx_tab; ++hp_round) {
    for (int retry = 0;; ++retry) {
        A.cap_rate = cap_rate;
        HIPCHK(hipMemsetAsync(ctx->stats.p, 0, NS_STATS_BYTES, st));
        // ---- plan: pieces, lengths of attempt 0, event capacity, visiting order ----
        HIPCHK(hipEventRecord(ctx->evt[1], st));
        k_nseg<<<grid_t, blk, 0, st>>>(A);
        HIPCHK(hipGetLastError());
        if ((rc = scan_u32(ctx, A.n_pieces, A.piece_off, n + 1))) return rc;
        // one piece per read unless the batch is chimeric (k_nseg): the total is n then, without a read-back (one stream round trip less
        // per worker call)
        if (prm->kind == NS_KIND_ALIGNED && prm->chimeric) {
            uint32_t tp32 = 0;
            if ((rc = read_small(ctx, st, &tp32, A.piece_off + n, 4))) return rc;
            tot_pieces = tp32;
        } else tot_pieces = n;
        if ((rc = ensure(ctx, ctx->pieces, (size_t)tot_pieces * sizeof(ns_piece) + 64))) return rc;
        A.pieces = (ns_piece *)ctx->pieces.p;
        A.list = nullptr; A.list_n = (uint32_t)n; A.attempt = 0;
        k_lengths<false><<<grid_t, blk, 0, st>>>(A);
        HIPCHK(hipGetLastError());
        if ((rc = scan_u64(ctx, A.ev_cap, A.ev_off, n + 1))) return rc;
        if ((rc = visiting_order(ctx, A.sort_key, A.sort_idx, n, list_a))) return rc;
        HIPCHK(hipEventRecord(ctx->evt[2], st));
        uint32_t n_multi = 0;                 // reads of several pieces: the head of the visiting order (visiting_order; 0 with NS_EXACT_ORDER)
        if (prm->chimeric && prm->kind == NS_KIND_ALIGNED && ctx->ord_bins.p && !getenv("NS_EXACT_ORDER")) {
            if ((rc = read_small(ctx, st, &tot_cap, A.ev_off + n, 8, &n_multi, (uint32_t *)ctx->ord_bins.p + 2 * NS_ORD_BINS, 4))) return rc;
        } else if ((rc = read_small(ctx, st, &tot_cap, A.ev_off + n, 8))) return rc;
        if ((rc = ensure(ctx, ctx->events, (size_t)tot_cap * sizeof(ns_event) + 64))) return rc;
        A.events = (ns_event *)ctx->events.p;
        // ---- passes: pass a generates attempt a of every read still without an accepted attempt ----
        uint32_t *cur = list_a, *nxt = list_b;
        uint32_t cur_n = (uint32_t)n;
        bool overflow = false;
        double ms_chain = 0;
        uint64_t used = tot_cap;                  // event slots handed out so far
        for (uint32_t a = 0;; ++a) {
            A.list = cur; A.list_n = cur_n; A.attempt = a; A.next_list = nxt; A.hole_at = 0; A.hole_len = 0; A.prio_thr = nullptr;
            A.l_off = nullptr; A.l_base = 0;
            if (a > 0) HIPCHK(hipMemsetAsync(A.next_n, 0, 4, st));      // (pass 0: the counters were zeroed as a whole at the top of the retry loop)
            const dim3 grid_p((cur_n + 255) / 256);
            A.p_need = nullptr; A.p_off = nullptr; A.p_base = 0;
            if (a > 0) {          // new lengths for the reads still open; their events go to a fresh region behind the earlier passes
                if (prm->kind == NS_KIND_ALIGNED && prm->chimeric) {     // ... and a new segment count with a new epoch (read_nseg)
                    if ((rc = ensure(ctx, ctx->p_need, ((size_t)cur_n + 1) * 4)) || (rc = ensure(ctx, ctx->p_off, ((size_t)cur_n + 1) * 4))) return rc;
                    k_replan<<<dim3((cur_n + 1 + 255) / 256), blk, 0, st>>>(A, (uint32_t *)ctx->p_need.p);
                    HIPCHK(hipGetLastError());
                    if ((rc = scan_u32(ctx, (const uint32_t *)ctx->p_need.p, (uint32_t *)ctx->p_off.p, (size_t)cur_n + 1))) return rc;
                    uint32_t extra = 0;
                    if ((rc = read_small(ctx, st, &extra, (uint32_t *)ctx->p_off.p + cur_n, 4))) return rc;
                    if (extra) {
                        if ((rc = ensure_keep(ctx, ctx->pieces, (size_t)(tot_pieces + extra) * sizeof(ns_piece) + 64, (size_t)tot_pieces * sizeof(ns_piece)))) return rc;
                        A.pieces = (ns_piece *)ctx->pieces.p;
                        A.p_need = (const uint32_t *)ctx->p_need.p; A.p_off = (const uint32_t *)ctx->p_off.p; A.p_base = (uint32_t)tot_pieces;
                        tot_pieces += extra;
                    }
                }
                if (cur_n <= 4096u) k_lengths<true><<<dim3((cur_n + 63) / 64), dim3(64), 0, st>>>(A);
                else k_lengths<false><<<grid_p, blk, 0, st>>>(A);
                HIPCHK(hipGetLastError());
                HIPCHK(hipMemsetAsync(A.l_cap + cur_n, 0, 8, st));
                if ((rc = scan_u64(ctx, A.l_cap, (uint64_t *)ctx->l_off.p, (size_t)cur_n + 1))) return rc;
                uint64_t pass_cap = 0;
                if ((rc = read_small(ctx, st, &pass_cap, (uint64_t *)ctx->l_off.p + cur_n, 8))) return rc;
                if ((rc = ensure_keep(ctx, ctx->events, (size_t)(used + pass_cap) * sizeof(ns_event) + 64, (size_t)used * sizeof(ns_event)))) return rc;
                A.events = (ns_event *)ctx->events.p;
                A.l_off = (const uint64_t *)ctx->l_off.p; A.l_base = used;
                used += pass_cap;
            }
            if (a == 0 && ctx->gate_wait) {                                // (bounded: the owner opens the gate on every way out of its call; an owner
                const auto t_end = std::chrono::steady_clock::now() + std::chrono::milliseconds(3);   // that blocks in front of its chain — a first-use
                while (!ctx->gate_wait->load(std::memory_order_acquire) && std::chrono::steady_clock::now() < t_end)   // hipMalloc, a result slot still
                    std::this_thread::yield();                                                         // crossing PCIe — is not waited for: 3 ms)
            }
            HIPCHK(hipEventRecord(ctx->evt[3], st));
            uint32_t n_coop = 0;
            if (prm->kind == NS_KIND_UNALIGNED)                           // its loop is a prefix sum (coop_unaligned_error_list); pass 0 visits the reads longest first
                n_coop = (a == 0 && lds && cur_n >= ctx->coop_min) ? std::max(cur_n >> ctx->ucoop_shift, 1u) : cur_n;
            else if (a == 0 && ctx->coop_ok && prm->kind == NS_KIND_ALIGNED && cur_n >= ctx->coop_min)    // longest 0.1 % (of the single-segment reads)
                n_coop = (cur_n - std::min(cur_n, n_multi)) >> ctx->coop_shift;
            const uint32_t coop_at = (a == 0 && prm->kind == NS_KIND_ALIGNED) ? std::min(cur_n, n_multi) : 0u;     // where that list starts in the visiting order
            // ... and the longest of the reads of several pieces (the head of the order): their chains are the sum of their pieces', so more of them
            // lie beyond the length at which a thread-per-read chain becomes the tail of the launch (NS_COOP_MULTI_SHIFT; same-box sweeps in profiles/r06/ab_chimeric_order.log)
            uint32_t m_coop = 0;
            if (coop_at && n_coop) {
                // (a thread per piece for the rest — below — leaves few of them too long for that side: 1/512; one thread per read: 1/32)
                const bool piece_threads = lds && !getenv("NS_NO_PIECE_THREADS");
                uint32_t sh = piece_threads ? (ctx->coop_shift > 1u ? ctx->coop_shift - 1u : 0u) : (ctx->coop_shift > 5u ? ctx->coop_shift - 5u : 0u);
                if (const char *d = getenv("NS_COOP_MULTI_SHIFT")) sh = (uint32_t)atoi(d) & 31u;
                m_coop = std::min(coop_at, std::max(coop_at >> sh, 64u));
            }
            if (n_coop) {      // wave-per-read for the head of the (length-sorted) list, thread-per-read for the rest
                GenArgs B = A; B.list_n = n_coop + m_coop; B.list = cur + (m_coop ? 0u : coop_at);
                if (m_coop) { B.hole_at = m_coop; B.hole_len = coop_at - m_coop; }      // [0, m_coop) and [coop_at, coop_at + n_coop) of the order
                { const char *d = getenv("NS_UCOOP_K"); B.coop_k1 = d && atoi(d) == 1 ? 1u : 0u; }
                HIPCHK(hipEventRecord(ctx->ev_fork, st));
                HIPCHK(hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
                // unaligned reads: the run-length tables in LDS (k_chain<true, true>; the image must fit next to nothing else: 64 KB)
                if (prm->kind == NS_KIND_UNALIGNED && ctx->lds_tables && ctx->ucoop_lds && (size_t)A.m.ct.n_words_mix * 8 <= 64u * 1024u)
                    k_chain<true, true><<<dim3(B.list_n), dim3(64), (size_t)A.m.ct.n_words_mix * 8, ctx->stream2>>>(B);
                else { B.coop_mix = (size_t)B.m.ct.n_words_mix * 8 <= 32u * 1024u ? 1u : 0u;
                       k_chain<false, true><<<dim3(B.list_n), dim3(64), B.coop_mix ? (size_t)B.m.ct.n_words_mix * 8 : 0, ctx->stream2>>>(B); }
                HIPCHK(hipGetLastError());
                HIPCHK(hipEventRecord(ctx->ev_join, ctx->stream2));
                if (coop_at) { A.list = cur + m_coop; A.hole_at = coop_at - m_coop; A.hole_len = n_coop; } else A.list = cur + n_coop;
                A.list_n = cur_n - n_coop - m_coop;
            }
            const uint32_t cb = lds ? ctx->chain_block : NS_CHAIN_BLOCK;
            // The reads of several pieces that stay on the thread-per-read side: a thread per PIECE (k_chain's piece mode 1: the error lists, each into
            // its piece's own share of the read's event slots), then a thread per read for what follows the lists (mode 2) — on a third stream, next
            // to the launch of the single-segment reads.  A thread that walks all pieces of its read has 2-4 times the trip count of any other, and
            // the slowest wavefront is what a chain launch waits for (profiles/r06/ab_chimeric_order.log).  NS_NO_PIECE_THREADS=1: one thread per read.
            uint32_t n_pt = 0;
            if (a == 0 && lds && n_coop && coop_at > m_coop && !getenv("NS_NO_PIECE_THREADS")) {
                if (!ctx->stream3) {
                    HIPCHK(hipStreamCreateWithFlags(&ctx->stream3, hipStreamNonBlocking));
                    HIPCHK(hipEventCreateWithFlags(&ctx->ev_join3, hipEventDisableTiming));
                }
                n_pt = coop_at - m_coop;
                GenArgs P = A; P.list = cur + m_coop; P.list_n = n_pt; P.hole_at = 0; P.hole_len = 0; P.prio_thr = nullptr;
                const size_t lds_p = ctx->lds_bytes + (P.ev_stage ? cb * 32u : 0u);
                HIPCHK(hipStreamWaitEvent(ctx->stream3, ctx->ev_fork, 0));
                P.piece_mode = 1;
                k_chain<true, false, true><<<dim3((unsigned)(((uint64_t)n_pt * NS_PIECE_SLOTS + cb - 1) / cb)), dim3(cb), lds_p, ctx->stream3>>>(P);
                P.piece_mode = 2;
                k_chain<true, false, true><<<dim3((n_pt + cb - 1) / cb), dim3(cb), lds_p, ctx->stream3>>>(P);
                HIPCHK(hipGetLastError());
                HIPCHK(hipEventRecord(ctx->ev_join3, ctx->stream3));
                A.list = cur + coop_at + n_coop; A.hole_at = 0; A.hole_len = 0; A.list_n = cur_n - coop_at - n_coop;
            }
            if (a == 0 && n_multi && !n_pt && prm->kind == NS_KIND_ALIGNED && !getenv("NS_PRIO_BY_POSITION")) A.prio_thr = (const uint32_t *)ctx->ord_bins.p + 2 * NS_ORD_BINS;
            const dim3 grid_c((A.list_n + cb - 1) / cb), blk_c(cb);
            if (!A.list_n) {}
            else if (lds) k_chain<true, false><<<grid_c, blk_c, ctx->lds_bytes + (A.ev_stage ? cb * 32u : 0u), st>>>(A);
            else k_chain<false, false><<<grid_c, blk_c, 0, st>>>(A);
            HIPCHK(hipGetLastError());
            if (n_coop) HIPCHK(hipStreamWaitEvent(st, ctx->ev_join, 0));
            if (n_pt) HIPCHK(hipStreamWaitEvent(st, ctx->ev_join3, 0));
            HIPCHK(hipEventRecord(ctx->evt[4], st));
            if (ctx->gate_signal) ctx->gate_signal->store(1, std::memory_order_release);
            fold_stats(ctx, st);
            if ((rc = read_small(ctx, st, stats, ctx->stats.p, sizeof stats))) return rc;
            HIPCHK(hipEventElapsedTime(&ms, ctx->evt[3], ctx->evt[4]));
            ms_chain += ms;
            if (stats[0] & NS_OVER_MASK) { overflow = true; break; }
            cur_n = (uint32_t)(stats[6] & 0xffffffffull);
            if (!cur_n) { tot_cap = used; break; }
            if (a + 1 >= NS_MAX_ATTEMPT)
                return fail(ctx, NS_EINVAL, "some reads found no acceptable length within the attempt limit "
                                            "(min_len/max_len too narrow for this model, or its reads do not fit the event record: runs <= 4095 bases, "
                                            "insertion / deletion balance within +-131071 bases per segment)");
            cur = nxt; nxt = cur == list_b ? list_c : list_b;      // (list_a keeps the length-sorted order of the batch for the record kernels)
        }
        info->ms_kernel[NS_K_EVENTS] = ms_chain;
#ifdef NS_CHAIN_CLOCK
        if (prm->kind == NS_KIND_ALIGNED) { unsigned long long c[16], z[16] = {0}; HIPCHK(hipMemcpyFromSymbol(c, HIP_SYMBOL(g_chain_clock), sizeof c)); HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_chain_clock), z, sizeof z));
          fprintf(stderr, "chain clock: lane 0 of the multi waves: aligned pieces %.3f ms, gaps %.3f ms, behind the lists %.3f ms (means) | ", c[6] ? c[8] * 1e-5 / c[6] : 0.0, c[6] ? c[9] * 1e-5 / c[6] : 0.0, c[6] ? c[10] * 1e-5 / c[6] : 0.0);
          fprintf(stderr, "chain clock: single waves %llu max %.3f ms mean %.3f ms (the group's last block: %llu) | multi waves %llu max %.3f ms mean %.3f ms (the group's last block: %llu) | ms_chain %.3f\n",
              c[2], c[0] * 1e-5, c[2] ? c[1] * 1e-5 / c[2] : 0.0, c[3] >> 32, c[6], c[4] * 1e-5, c[6] ? c[5] * 1e-5 / c[6] : 0.0, c[7] >> 32, ms_chain); }
#endif
        if (!overflow) break;
        info->n_overflow += stats[0] & NS_OVER_MASK;
        if (retry >= 6) return fail(ctx, NS_ENOMEM, "event capacity overflow persists after 6 retries");
        cap_rate *= 2.0; A.cap_gap_mul *= 2;          // rare: more events per base than planned -> re-plan the batch with twice the rates
    }
    if (ir_on && (rc = ir_splice())) return rc;
    if (!A.hp) break;
    if ((rc = hp_stage1(ctx, prm, A, n, tot_pieces, tot_cap, stats, &ms_hp))) return rc;
    if (!stats[5]) break;
    // some reads failed the final length check (S:1429): they advanced their attempt state; every other read restarts at
    // its accepted attempt, so re-running the batch reproduces them bit for bit
    if (hp_round >= (int)NS_MAX_ATTEMPT) return fail(ctx, NS_EINVAL, "reads keep failing the final length check in -k mode");
    A.keep_state = 1;
    }
    if (A.errlen_later && !A.hp && !meta_al) {   // (-k: k_hp_filter_w has computed the sizes of the rows that survive the filter; metagenome: per pass)
        k_errlen<<<dim3((unsigned)((n + NS_WPB - 1) / NS_WPB)), dim3(64 * NS_WPB), 0, st>>>(A);
        HIPCHK(hipGetLastError());
    }
    if ((rc = scan_u64(ctx, A.rec_len, A.rec_off, n + 1))) return rc;
    if (prm->emit_errlog && (rc = scan_u64(ctx, A.err_len, A.err_off, n + 1))) return rc;
    if ((rc = prm->emit_errlog ? read_small(ctx, st, &info->record_bytes, A.rec_off + n, 8, &info->errlog_bytes, A.err_off + n, 8)
                               : read_small(ctx, st, &info->record_bytes, A.rec_off + n, 8))) return rc;
    // result slot of this batch: the other one while the last batch is still being copied out (ns_sink_write); a slot is reused
    // once its copies have left the device
    int slot = ctx->slot;
    if (prm->emit_records == 1u && ctx->io) {
        if (ctx->io->slot_busy(slot)) slot ^= 1;
        ctx->io->wait_slot(slot);
    }
    if (prm->emit_records == 1u && ((rc = ensure(ctx, ctx->rec_slot[slot], (size_t)info->record_bytes + 64)) ||
                                    (rc = ensure(ctx, ctx->err_slot[slot], (size_t)info->errlog_bytes + 64))))
        return rc;
    if (prm->emit_records == 0) info->errlog_bytes = 0;       // (no records: no error-profile image either; NS_EMIT_SIZES keeps the size)
    A.records = (uint8_t *)ctx->rec_slot[slot].p; A.errlog = (uint8_t *)ctx->err_slot[slot].p;
    A.cls = nullptr;
    if (prm->emit_records == 1u && prm->fastq && prm->kind != NS_KIND_UNALIGNED) {      // class words: k_materialise -> k_qualities (cls_word0)
        const size_t per_read = prm->chimeric ? 2u * (2u * NS_MAX_SEG - 1u) : 2u;          // cls_per_read
        if ((rc = ensure(ctx, ctx->cls, (((size_t)info->record_bytes >> 4) + per_read * (n + 1) + 64) * 4))) return rc;
        A.cls = (uint32_t *)ctx->cls.p;
    }
    const uint64_t max_unaligned = prm->kind == NS_KIND_UNALIGNED ? stats[1] : 0;      // emitted bases of the batch (k_chain): bounds the dense kernel's grid
    HIPCHK(hipEventRecord(ctx->evt[5], st));
    const bool write_rec = prm->emit_records == 1u;            // (2 = NS_EMIT_SIZES: the sizes of the images only)
    const bool side_names = write_rec;                        // names + framing on the second stream, next to the record kernel (round 5: also
                                                              // next to the second record pass of -k: 0.37 ms per 950 000 reads on the main stream)
    if (side_names) {
        HIPCHK(hipEventRecord(ctx->ev_fork, st));
        HIPCHK(hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
    }
    k_names<<<grid_t, blk, 0, side_names ? ctx->stream2 : st>>>(A);
    HIPCHK(hipGetLastError());
    if (side_names) HIPCHK(hipEventRecord(ctx->ev_join, ctx->stream2));
    HIPCHK(hipEventRecord(ctx->evt[6], st));
    if (A.hp) {          // second record pass of -k: the scratch read + its homopolymer edits -> the record
        if (write_rec && (rc = launch_materialise(ctx, A, n, prm->fastq != 0, tot_cap, nullptr, (meta_al || trx_tab) ? nullptr : list_a, MAT_HP_FINAL))) return rc;
        k_hp_report<<<dim3((unsigned)((tot_pieces + 255) / 256)), blk, 0, st>>>(A, tot_pieces);      // the pieces report their emitted length,
        HIPCHK(hipGetLastError());                                                                    // like the path without -k
    } else if (write_rec) {
        // (k_names runs NEXT to the record kernels on the second stream: they write different bytes of the image)
        if ((rc = launch_materialise(ctx, A, n, prm->fastq != 0, tot_cap, nullptr, (meta_al || trx_tab) ? nullptr : list_a, MAT_REF, max_unaligned))) return rc;
    }
    if (side_names) HIPCHK(hipStreamWaitEvent(st, ctx->ev_join, 0));
    HIPCHK(hipEventRecord(ctx->evt[7], st));
    if (prm->emit_errlog && write_rec) {
        // the block buffer of the kernel: the small one (six wavefronts per SIMD instead of four) when 64 average rows of this batch fit it
        const uint64_t rows = stats[3] ? stats[3] : 1;
        const bool small_buf = !getenv("NS_ERRLOG_BUF_LARGE") && (info->errlog_bytes / rows + 5) * 64 <= NS_ERR_BUF_SMALL;
        if (small_buf) k_errlog<NS_ERR_BUF_SMALL><<<dim3((unsigned)n), dim3(64), 0, st>>>(A);
        else k_errlog<NS_ERR_BUF_LARGE><<<dim3((unsigned)n), dim3(64), 0, st>>>(A);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(ctx->evt[8], st));
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipEventElapsedTime(&ms, ctx->evt[0], ctx->evt[8])); info->ms_total = ms;
    HIPCHK(hipEventElapsedTime(&ms, ctx->evt[1], ctx->evt[2])); info->ms_kernel[NS_K_LENGTHS] = ms;   // nseg + lengths + scans + sort
    HIPCHK(hipEventElapsedTime(&ms, ctx->evt[5], ctx->evt[6])); info->ms_kernel[NS_K_SCAN] = ms;      // names/framing
    HIPCHK(hipEventElapsedTime(&ms, ctx->evt[6], ctx->evt[7])); info->ms_kernel[NS_K_MATERIALISE] = ms;
    HIPCHK(hipEventElapsedTime(&ms, ctx->evt[7], ctx->evt[8])); info->ms_kernel[NS_K_ERRLOG] = ms;
    if (ctx->rec_timed) { HIPCHK(hipEventElapsedTime(&ms, ctx->evt[12], ctx->evt[13])); info->ms_kernel[NS_K_RECORD_KERNEL] = ms; }
    info->ms_kernel[NS_K_HP] = ms_hp;
    info->n_reads = n; info->n_pieces = tot_pieces; info->n_events = tot_cap;
    info->total_bases = stats[1]; info->total_ref_bases = stats[2]; info->events_used = stats[3];
    info->n_range_redraws = stats[0] >> 40;
    info->spliced_bytes = ctx->spliced_bytes;
    ctx->last = *info;
    ctx->last.n_events = tot_cap;
    if (prm->emit_records != 1u) { ctx->last.record_bytes = 0; ctx->last.errlog_bytes = 0; }      // nothing to copy out
    ctx->slot = slot;
    ctx->has_batch = true;
    return NS_OK;
}

// ---- ns_generate_step: the aligned and the unaligned worker call of one step side by side (include/nanosim_amd.h) ----
// what the companion borrows from its owner: everything ns_generate reads that ns_set_* / ns_load_model fill (device pointers by value)
static void lend_tables(const ns_ctx *ctx, ns_ctx *c) {
    c->m = ctx->m; c->ref = ctx->ref; c->has_model = ctx->has_model; c->has_ref = ctx->has_ref;
    c->cap_rate = ctx->cap_rate; c->ref_nbases = ctx->ref_nbases;
    c->lds_tables = ctx->lds_tables; c->lds_bytes = ctx->lds_bytes; c->coop_ok = ctx->coop_ok; c->chain_block = ctx->chain_block;
    c->nspecies = ctx->nspecies; c->species_chrom_off = ctx->species_chrom_off;      // (DevBuf by value: not freed by the companion)
    c->tx = ctx->tx; c->has_trx = ctx->has_trx;
    c->has_abun = false; c->has_inflated = false; c->has_ir = false;                   // (aligned workers only: S:814-1040, 1156-1192)
}
static void step_worker_main(ns_ctx *owner) {
    ns_ctx::StepWorker *w = owner->step;
    std::unique_lock<std::mutex> lk(w->mu);
    for (;;) {
        w->cv.wait(lk, [w] { return w->quit || (w->prm && !w->done); });
        if (w->quit) return;
        const ns_params *prm = w->prm; ns_batch_info *info = w->info;
        lk.unlock();
        const int rc = ns_generate(owner->companion, prm, info);
        lk.lock();
        w->rc = rc; w->done = true;
        w->cv.notify_all();
    }
}
int ns_step_context(ns_ctx *ctx, ns_ctx **out) {
    if (!ctx) return NS_EINVAL;
    if (!out) return fail(ctx, NS_EINVAL, "null output pointer");
    *out = nullptr;
    if (ctx->borrowed) return fail(ctx, NS_EINVAL, "a step companion has no companion of its own");
    if (!ctx->companion) {
        ns_ctx *c = nullptr;
        const char *pe = getenv("NS_STEP_PRIO");
        const int rc = create_ctx(ctx->device, &c, pe ? atoi(pe) : 0);
        if (rc) return fail(ctx, rc, "ns_generate_step: the companion context could not be created");
        c->borrowed = true; c->owner = ctx;
        ns_set_background(c, 1);
        ctx->companion = c;
    }
    lend_tables(ctx, ctx->companion);
    *out = ctx->companion;
    return NS_OK;
}
int ns_generate_step(ns_ctx *ctx, const ns_params *aligned, const ns_params *unaligned, ns_batch_info info[2]) {
    if (!ctx) return NS_EINVAL;
    if (!info) return fail(ctx, NS_EINVAL, "null info");
    if (ctx->borrowed) return fail(ctx, NS_EINVAL, "ns_generate_step on a step companion");
    if (aligned && aligned->kind == NS_KIND_UNALIGNED) return fail(ctx, NS_EINVAL, "ns_generate_step: the first call is the aligned (or perfect) worker call");
    if (unaligned && unaligned->kind != NS_KIND_UNALIGNED) return fail(ctx, NS_EINVAL, "ns_generate_step: the second call is the unaligned worker call");
    memset(info, 0, 2 * sizeof *info);
    if (!unaligned) return aligned ? ns_generate(ctx, aligned, &info[0]) : NS_OK;
    ns_ctx *c = nullptr;
    int rc = ns_step_context(ctx, &c);                        // (creates the companion on first use; lends it the tables as they are now)
    if (rc) return rc;
    if (!aligned) {
        rc = ns_generate(c, unaligned, &info[1]);
        if (rc) ctx->err = "unaligned worker call: " + c->err;
        return rc;
    }
    if (!ctx->step) {
        ctx->step = new ns_ctx::StepWorker();
        ctx->step->th = std::thread(step_worker_main, ctx);
    }
    ns_ctx::StepWorker *w = ctx->step;
    const char *gate_env = getenv("NS_STEP_GATE");
    const bool gated = !(gate_env && gate_env[0] == '0') && !aligned->meta && !aligned->trx;
    ctx->gate.store(0); ctx->gate_signal = gated ? &ctx->gate : nullptr; c->gate_wait = gated ? &ctx->gate : nullptr;
    { std::lock_guard<std::mutex> lk(w->mu); w->prm = unaligned; w->info = &info[1]; w->done = false; w->rc = 0; }
    w->cv.notify_all();
    const int rc_al = ns_generate(ctx, aligned, &info[0]);
    ctx->gate.store(1, std::memory_order_release); ctx->gate_signal = nullptr;
    int rc_un;
    { std::unique_lock<std::mutex> lk(w->mu); w->cv.wait(lk, [w] { return w->done; }); rc_un = w->rc; w->prm = nullptr; w->info = nullptr; }
    c->gate_wait = nullptr;
    if (rc_al) return rc_al;
    if (rc_un) { ctx->err = "unaligned worker call: " + c->err; return rc_un; }
    return NS_OK;
}

static int result_buf(ns_ctx *ctx, int which, const void **p, uint64_t *size) {
    const ns_batch_info &b = ctx->last;
    switch (which) {
        case NS_BUF_RECORDS: *p = ctx->rec_slot[ctx->slot].p; *size = b.record_bytes; return NS_OK;
        case NS_BUF_READS: *p = ctx->reads.p; *size = b.n_reads * sizeof(ns_read); return NS_OK;
        case NS_BUF_PIECES: *p = ctx->pieces.p; *size = b.n_pieces * sizeof(ns_piece); return NS_OK;
        case NS_BUF_EVENTS: *p = ctx->events.p; *size = b.n_events * sizeof(ns_event); return NS_OK;
        case NS_BUF_ERRLOG: *p = ctx->err_slot[ctx->slot].p; *size = b.errlog_bytes; return NS_OK;
        case NS_BUF_POLYA: *p = ctx->polya.p; *size = ctx->polya.p ? b.n_reads * 2 : 0; return NS_OK;
        case NS_BUF_SPLICED: *p = ctx->spliced.p; *size = ctx->spliced_bytes; return NS_OK;
        default: return NS_EINVAL;
    }
}

int ns_copy_out(ns_ctx *ctx, int which, void *host_dst, uint64_t offset, uint64_t nbytes) {
    if (!ctx) return NS_EINVAL;
    if (!ctx->has_batch) return fail(ctx, NS_ESTATE, "no batch to copy");
    const void *p; uint64_t size;
    if (result_buf(ctx, which, &p, &size)) return fail(ctx, NS_EINVAL, "unknown buffer id");
    if (offset > size || nbytes > size - offset) return fail(ctx, NS_EINVAL, "copy range exceeds buffer");
    if (!nbytes) return NS_OK;
    if (!host_dst) return fail(ctx, NS_EINVAL, "null destination");
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipMemcpyAsync(host_dst, static_cast<const uint8_t *>(p) + offset, nbytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return NS_OK;
}

int ns_host_alloc(ns_ctx *ctx, uint64_t nbytes, void **out) {
    if (!ctx) return NS_EINVAL;
    if (!out || !nbytes) return fail(ctx, NS_EINVAL, "ns_host_alloc: null destination / zero size");
    *out = nullptr;
    HIPCHK(hipSetDevice(ctx->device));
    hipError_t e = hipHostMalloc(out, nbytes, hipHostMallocDefault);
    if (e != hipSuccess) { *out = nullptr; return fail(ctx, NS_ENOMEM, std::string("hipHostMalloc: ") + hipGetErrorString(e)); }
    return NS_OK;
}

int ns_host_free(ns_ctx *ctx, void *p) {
    if (!ctx) return NS_EINVAL;
    if (!p) return NS_OK;
    HIPCHK(hipHostFree(p));
    return NS_OK;
}

// ---- output sinks (ns_io.h): the worker's out_reads.write / out_error.write (S:1437-1443, 2006-2008) as an asynchronous pipeline ----
int ns_sink_open(ns_ctx *ctx, int fd, uint64_t file_off, ns_sink **out) {
    if (!ctx) return NS_EINVAL;
    if (!out) return fail(ctx, NS_EINVAL, "ns_sink_open: null destination");
    *out = nullptr;
    if (!ctx->io) {
        IoEngine *io = new IoEngine();
        std::string msg;
        if (io->start(ctx->device, msg)) { io->shutdown(); delete io; return fail(ctx, NS_ENOMEM, msg); }
        ctx->io = io;
    }
    ns_sink *s = new ns_sink();
    s->fd = fd; s->off = file_off;
    ctx->sinks.push_back(s);
    *out = s;
    return NS_OK;
}

static bool own_sink(ns_ctx *ctx, ns_sink *s) { return s && std::find(ctx->sinks.begin(), ctx->sinks.end(), s) != ctx->sinks.end(); }

int ns_sink_put(ns_ctx *ctx, ns_sink *s, const void *host, uint64_t n) {
    if (!ctx) return NS_EINVAL;
    if (!own_sink(ctx, s)) return fail(ctx, NS_EINVAL, "unknown sink");
    if (n && !host) return fail(ctx, NS_EINVAL, "ns_sink_put: null source");
    uint64_t done = 0;
    while (s->fd >= 0 && done < n) {
        const ssize_t w = pwrite(s->fd, static_cast<const uint8_t *>(host) + done, (size_t)(n - done), (off_t)(s->off + done));
        if (w < 0 && errno == EINTR) continue;
        if (w <= 0) return fail(ctx, NS_EIO, std::string("write: ") + strerror(w < 0 ? errno : ENOSPC));
        done += (uint64_t)w;
    }
    s->off += n; s->queued += n; s->written += n;
    return NS_OK;
}

int ns_sink_write_range(ns_ctx *ctx, ns_sink *s, int which, uint64_t offset, uint64_t nbytes) {
    if (!ctx) return NS_EINVAL;
    if (!own_sink(ctx, s)) return fail(ctx, NS_EINVAL, "unknown sink");
    if (!ctx->has_batch) return fail(ctx, NS_ESTATE, "no batch to write");
    if (which != NS_BUF_RECORDS && which != NS_BUF_ERRLOG) return fail(ctx, NS_EINVAL, "ns_sink_write: records or error profile only");
    const void *p; uint64_t size;
    if (result_buf(ctx, which, &p, &size)) return fail(ctx, NS_EINVAL, "unknown buffer id");
    if (offset > size || nbytes > size - offset) return fail(ctx, NS_EINVAL, "ns_sink_write_range: range exceeds the buffer");
    ctx->io->enqueue(s, static_cast<const uint8_t *>(p) + offset, nbytes, ctx->slot);
    return NS_OK;
}

int ns_sink_write(ns_ctx *ctx, ns_sink *s, int which) {
    if (!ctx) return NS_EINVAL;
    if (!ctx->has_batch) return fail(ctx, NS_ESTATE, "no batch to write");
    const void *p; uint64_t size;
    if (result_buf(ctx, which, &p, &size)) return fail(ctx, NS_EINVAL, "unknown buffer id");
    return ns_sink_write_range(ctx, s, which, 0, size);
}

int ns_record_offsets(ns_ctx *ctx, const uint64_t *read_index, uint32_t n, uint64_t *rec_off, uint64_t *err_off) {
    if (!ctx) return NS_EINVAL;
    if (!ctx->has_batch) return fail(ctx, NS_ESTATE, "no batch");
    if (n && (!read_index || !rec_off)) return fail(ctx, NS_EINVAL, "ns_record_offsets: null argument");
    const uint64_t nr = ctx->last.n_reads;
    for (uint32_t i = 0; i < n; ++i) if (read_index[i] > nr) return fail(ctx, NS_EINVAL, "ns_record_offsets: read index beyond the batch");
    if (!nr || !ctx->last.record_bytes) {              // empty batch / no record image
        for (uint32_t i = 0; i < n; ++i) { rec_off[i] = 0; if (err_off) err_off[i] = 0; }
        return NS_OK;
    }
    HIPCHK(hipSetDevice(ctx->device));
    const bool with_err = err_off && ctx->last.errlog_bytes;
    uint64_t *slots = reinterpret_cast<uint64_t *>(ctx->pin_small);            // 1 KB of page-locked memory: 64 + 64 values per round
    for (uint32_t i0 = 0; i0 < n; i0 += 64) {
        const uint32_t m = std::min<uint32_t>(64u, n - i0);
        for (uint32_t i = 0; i < m; ++i) {
            HIPCHK(hipMemcpyAsync(slots + i, (const uint64_t *)ctx->rec_off.p + read_index[i0 + i], 8, hipMemcpyDeviceToHost, ctx->stream));
            if (with_err) HIPCHK(hipMemcpyAsync(slots + 64 + i, (const uint64_t *)ctx->err_off.p + read_index[i0 + i], 8, hipMemcpyDeviceToHost, ctx->stream));
        }
        HIPCHK(hipStreamSynchronize(ctx->stream));
        for (uint32_t i = 0; i < m; ++i) { rec_off[i0 + i] = slots[i]; if (err_off) err_off[i0 + i] = with_err ? slots[64 + i] : 0; }
    }
    return NS_OK;
}

int ns_sink_drain(ns_ctx *ctx, ns_sink *s, uint64_t *file_off) {
    if (!ctx) return NS_EINVAL;
    if (!own_sink(ctx, s)) return fail(ctx, NS_EINVAL, "unknown sink");
    ctx->io->wait_sink(s);
    if (file_off) *file_off = s->off;
    { std::lock_guard<std::mutex> g(ctx->io->mu); if (!ctx->io->err.empty()) return fail(ctx, NS_EHIP, ctx->io->err); }
    if (const int e = s->err.load()) return fail(ctx, NS_EIO, std::string("write: ") + strerror(e));
    return NS_OK;
}

int ns_sink_close(ns_ctx *ctx, ns_sink *s) {
    if (!ctx) return NS_EINVAL;
    if (!s) return NS_OK;
    const int rc = ns_sink_drain(ctx, s, nullptr);
    if (rc == NS_EINVAL) return rc;
    ctx->sinks.erase(std::find(ctx->sinks.begin(), ctx->sinks.end(), s));
    delete s;
    return rc;
}

int ns_io_counters(ns_ctx *ctx, ns_io_stats *out, int reset) {
    if (!ctx) return NS_EINVAL;
    if (!out) return fail(ctx, NS_EINVAL, "null destination");
    memset(out, 0, sizeof *out);
    if (!ctx->io) return NS_OK;
    std::lock_guard<std::mutex> g(ctx->io->mu);
    out->bytes = ctx->io->bytes; out->dma_ms = ctx->io->dma_ms; out->wait_staging_s = ctx->io->wait_free_s; out->write_s = ctx->io->write_s;
    out->slice_bytes = ctx->io->slice_bytes; out->n_slices = (uint32_t)ctx->io->slices.size(); out->n_threads = (uint32_t)ctx->io->writers.size();
    if (reset) { ctx->io->bytes = 0; ctx->io->dma_ms = ctx->io->wait_free_s = ctx->io->write_s = 0; }
    return NS_OK;
}

// the characterisation stage's counting loop (include/nanosim_amd.h: ns_cs_hist; src/besthit_to_histogram.py:316-365)
static int histograms(ns_ctx *ctx, const uint8_t *cs, const uint8_t *qry, bool maf, uint64_t nbytes, const uint64_t *aln_off, uint32_t n_aln, ns_cs_hist *h);
int ns_cs_histograms(ns_ctx *ctx, const uint8_t *cs, uint64_t nbytes, const uint64_t *aln_off, uint32_t n_aln, ns_cs_hist *h) {
    return histograms(ctx, cs, nullptr, false, nbytes, aln_off, n_aln, h);
}
int ns_maf_histograms(ns_ctx *ctx, const uint8_t *ref_lines, const uint8_t *query_lines, uint64_t nbytes, const uint64_t *aln_off, uint32_t n_aln, ns_cs_hist *h) {
    if (ctx && n_aln && nbytes && !query_lines) return fail(ctx, NS_EINVAL, "ns_maf_histograms: null argument");
    return histograms(ctx, ref_lines, query_lines, true, nbytes, aln_off, n_aln, h);
}
static int histograms(ns_ctx *ctx, const uint8_t *cs, const uint8_t *qry, bool maf, uint64_t nbytes, const uint64_t *aln_off, uint32_t n_aln, ns_cs_hist *h) {
    if (!ctx) return NS_EINVAL;
    if (!h || (n_aln && (!aln_off || (!cs && nbytes)))) return fail(ctx, NS_EINVAL, "ns_cs_histograms: null argument");
    for (uint32_t a = 0; a < n_aln; ++a)
        if (aln_off[a] > aln_off[a + 1] || aln_off[a + 1] > nbytes) return fail(ctx, NS_EINVAL, "ns_cs_histograms: offsets not ascending / beyond the strings");
    const uint32_t cap = h->cap_match2d;
    if (h->match_list && (!cap || cap > 65536u)) return fail(ctx, NS_EINVAL, "ns_cs_histograms: cap_match2d must be 1 .. 65536");
    HIPCHK(hipSetDevice(ctx->device));
    uint64_t *m2_host = h->match_list;
    memset(h->dic, 0, sizeof h->dic); memset(h->error_list, 0, sizeof h->error_list); memset(h->first_error, 0, sizeof h->first_error);
    h->max_match = h->n_match2d_overflow = h->n_skip = 0; h->ms_kernel = 0;
    if (!n_aln) { if (m2_host) memset(m2_host, 0, (size_t)cap * cap * 8); return NS_OK; }
    const size_t n_small = 5 * 1001 + 24 + 8;
    void *d_cs = nullptr, *d_off = nullptr, *d_small = nullptr, *d_m2 = nullptr, *d_key = nullptr, *d_tmp = nullptr, *d_qry = nullptr;
    auto release = [&]() { for (void *p : {d_cs, d_off, d_small, d_m2, d_key, d_tmp, d_qry}) if (p) { hipError_t e = hipFree(p); (void)e; } };
    hipError_t e = hipMalloc(&d_cs, (size_t)nbytes + 16);
    if (e == hipSuccess && maf) e = hipMalloc(&d_qry, (size_t)nbytes + 16);
    if (e == hipSuccess) e = hipMalloc(&d_off, ((size_t)n_aln + 1) * 8);
    if (e == hipSuccess) e = hipMalloc(&d_small, n_small * 8);
    if (e == hipSuccess && m2_host) e = hipMalloc(&d_m2, (size_t)cap * cap * 8);
    if (e != hipSuccess) { release(); (void)hipGetLastError(); return fail(ctx, NS_ENOMEM, std::string("ns_cs_histograms: hipMalloc: ") + hipGetErrorString(e)); }
    hipStream_t st = ctx->stream;
    e = hipMemcpyAsync(d_cs, cs, (size_t)nbytes, hipMemcpyHostToDevice, st);
    if (e == hipSuccess && maf && nbytes) e = hipMemcpyAsync(d_qry, qry, (size_t)nbytes, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d_off, aln_off, ((size_t)n_aln + 1) * 8, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemsetAsync(d_small, 0, n_small * 8, st);
    if (e == hipSuccess && d_m2) e = hipMemsetAsync(d_m2, 0, (size_t)cap * cap * 8, st);
    if (e == hipSuccess) e = hipEventRecord(ctx->evt[14], st);
    // Alignments of 1 kb .. 100 kb on neighbouring lanes diverge like the thread-per-read chain did before its length sort: the walks are
    // visited by descending length of their cs strings (the counts are sums: any order gives the same tables).  NS_CS_NO_SORT=1: file order.
    uint32_t *d_order = nullptr;
    if (e == hipSuccess && n_aln > 64 && !getenv("NS_CS_NO_SORT")) {
        size_t tmp = 0;
        e = hipMalloc(&d_key, (size_t)n_aln * 16);                 // key, index, sorted key, sorted index
        if (e == hipSuccess) {
            uint32_t *key = (uint32_t *)d_key, *idx = key + n_aln, *key2 = idx + n_aln, *idx2 = key2 + n_aln;
            k_cs_len<<<dim3((n_aln + 255u) / 256u), dim3(256), 0, st>>>((const uint64_t *)d_off, n_aln, key, idx);
            e = hipGetLastError();
            if (e == hipSuccess) e = hipcub::DeviceRadixSort::SortPairsDescending(nullptr, tmp, key, key2, idx, idx2, (int)n_aln, 0, 32, st);
            if (e == hipSuccess) e = hipMalloc(&d_tmp, tmp + 16);
            if (e == hipSuccess) e = hipcub::DeviceRadixSort::SortPairsDescending(d_tmp, tmp, key, key2, idx, idx2, (int)n_aln, 0, 32, st);
            d_order = idx2;
        }
    }
    if (e == hipSuccess) {
        CsHistDev H;
        H.dic = (unsigned long long *)d_small; H.err = H.dic + 5 * 1001; H.misc = H.err + 24;
        H.m2 = (unsigned long long *)d_m2; H.cap2 = cap;
        k_cs_hist<<<dim3((n_aln + 255u) / 256u), dim3(256), 0, st>>>((const uint8_t *)d_cs, (const uint64_t *)d_off, n_aln, H, d_order, (const uint8_t *)d_qry);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipEventRecord(ctx->evt[15], st);
    std::vector<unsigned long long> small(n_small);
    if (e == hipSuccess) e = hipMemcpyAsync(small.data(), d_small, n_small * 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && d_m2) e = hipMemcpyAsync(m2_host, d_m2, (size_t)cap * cap * 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    float ms = 0;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, ctx->evt[14], ctx->evt[15]);
    release();
    if (e != hipSuccess) return fail(ctx, NS_EHIP, std::string("ns_cs_histograms: ") + hipGetErrorString(e));
    for (int w = 0; w < 5; ++w) for (int v = 0; v <= 1000; ++v) h->dic[w][v] = small[(size_t)w * 1001 + v];
    for (int i = 0; i < 18; ++i) h->error_list[i] = small[5 * 1001 + i];
    for (int i = 0; i < 3; ++i) h->first_error[i] = small[5 * 1001 + 18 + i];
    h->max_match = small[5 * 1001 + 24]; h->n_match2d_overflow = small[5 * 1001 + 25]; h->n_skip = small[5 * 1001 + 26];
    h->ms_kernel = ms;
    return NS_OK;
}

const void *ns_device_ptr(ns_ctx *ctx, int which) {
    if (!ctx || !ctx->has_batch) return nullptr;
    const void *p; uint64_t size;
    if (result_buf(ctx, which, &p, &size)) return nullptr;
    return p;
}

}  // extern "C"
