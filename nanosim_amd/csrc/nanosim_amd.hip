// nanosim_amd.hip — gfx950 kernels + C-ABI host layer of the read-generation engine.
//
// Pipeline of one ns_generate() call (one worker call of the reference, S:1266-1454 / S:1482-1549):
//   k_plan        thread/read   segment count + event capacity                       (S:1276-1299)
//   scan          rocPRIM       piece / event offsets
//   k_events      thread/read   lengths, strand, error_list Markov chains, acceptance, positions
//                                                                                     (S:1283-1402, 1833-1916, 1784-1830, 1694-1781)
//   scan          rocPRIM       record / error-log offsets
//   k_names       thread/read   ">name\n", "+\n" framing                              (S:1390-1402, 1437-1443)
//   k_materialise wave/read     case_convert + mutate_read + head/tail + revcomp (+ qualities)
//                                                                                     (S:743-755, 1919-2015, 1421-1435)
//   k_errlog      wave/read     _aligned_error_profile rows                           (S:2006-2008)
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>

#include "ns_device.h"
#include "ns_materialise.h"

// ---------------------------------------------------------------------------------------------------------
// kernel arguments
// ---------------------------------------------------------------------------------------------------------
struct GenArgs {
    ns_params prm;
    DevModel m;
    DevRef ref;
    double cap_rate;            // event capacity per aligned reference base
    uint32_t cap_gap_mul;       // event capacity per gap/unaligned reference base
    // per-read planning arrays (n+1)
    uint32_t *n_pieces;         // in: count, after scan: offsets (separate array piece_off)
    uint32_t *piece_off;
    uint64_t *ev_cap;
    uint64_t *ev_off;
    uint64_t *rec_len;
    uint64_t *rec_off;
    uint64_t *err_len;
    uint64_t *err_off;
    uint16_t *name_len;
    // results
    ns_read *reads;
    ns_piece *pieces;
    ns_event *events;
    uint8_t *records;
    uint8_t *errlog;
    unsigned long long *stats;  // [0] overflow reads [1] total bases [2] total ref bases [3] events [4] failed reads
};

__device__ __forceinline__ ns_key make_key(const ns_params &prm, uint64_t r) {
    uint64_t g = prm.first_read + r;
    return ns_key{(uint32_t)prm.seed, (uint32_t)(prm.seed >> 32), (uint32_t)g, (uint32_t)(g >> 32)};
}

__device__ __forceinline__ uint32_t read_nseg(const GenArgs &A, const ns_key &key) {
    if (A.prm.kind != NS_KIND_ALIGNED || !A.prm.chimeric) return 1;
    u32x4 w = ns_draw(key, ST_NSEG, 0, 0, 0, 0);                                     // S:1276-1277
    uint32_t nseg = (uint32_t)table_value(A.m.nseg_cdf, A.m.nseg_n, u32_to_p(w.x));
    return nseg > NS_MAX_SEG ? NS_MAX_SEG : nseg;
}

// ---------------------------------------------------------------------------------------------------------
// k_plan: pieces per read and event capacity from the epoch-0 lengths
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_plan(GenArgs A) {
    uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > A.prm.n_reads) return;
    if (r == A.prm.n_reads) { A.n_pieces[r] = 0; A.ev_cap[r] = 0; return; }
    ns_key key = make_key(A.prm, r);
    uint64_t cap = 0;
    uint32_t np = 1;
    if (A.prm.kind == NS_KIND_UNALIGNED) {
        int64_t l = unaligned_length(A.m, A.prm, key, 0);
        if (l < 0) l = 0;
        cap = (uint64_t)l * A.cap_gap_mul + 64;
    } else if (A.prm.kind == NS_KIND_PERFECT) {
        cap = 0;
    } else {
        uint32_t nseg = read_nseg(A, key);
        np = 2 * nseg - 1;
        for (uint32_t s = 0; s < nseg; ++s) {
            int64_t l = 0;
            if (!seg_length(A.m, A.prm, key, s, 0, l)) l = 0;
            cap += (uint64_t)((double)l * A.cap_rate) + 64;
        }
        for (uint32_t g = 0; g + 1 < nseg; ++g) cap += (uint64_t)gap_length(A.m, key, g, 0) * A.cap_gap_mul + 64;
    }
    A.n_pieces[r] = np;
    A.ev_cap[r] = cap;
}

// ---------------------------------------------------------------------------------------------------------
// k_events: one thread per read; the whole accept/reject loop of S:1283-1449 with per-read retry counters
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_events(GenArgs A) {
    uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > A.prm.n_reads) return;
    if (r == A.prm.n_reads) { A.rec_len[r] = 0; A.err_len[r] = 0; return; }
    const ns_params &prm = A.prm;
    const int kind = (int)prm.kind;
    ns_key key = make_key(prm, r);
    const uint32_t nseg = read_nseg(A, key);
    const uint32_t n_pieces = (kind == NS_KIND_ALIGNED) ? 2 * nseg - 1 : 1;
    ns_piece *pc = A.pieces + A.piece_off[r];
    ns_event *ev_base = A.events + A.ev_off[r];
    const uint64_t ev_cap64 = A.ev_off[r + 1] - A.ev_off[r];
    const uint32_t ev_cap = ev_cap64 > 0xffffffffull ? 0xffffffffu : (uint32_t)ev_cap64;

    ns_read rd;
    rd.rec_off = 0; rd.piece_off = A.piece_off[r]; rd.n_pieces = (uint16_t)n_pieces; rd.reversed = 0; rd.flags = 1;
    rd.head = rd.tail = rd.seq_len = 0; rd.attempts = 0;
    uint32_t name_len = 0;
    uint64_t err_len = 0;
    bool done = false, overflow = false;
    uint32_t epoch = 0, fails = 0;
    for (uint32_t a = 0; a < NS_MAX_ATTEMPT && !done; ++a) {
        bool ok = true;
        // lengths are pure functions of (read, seg, epoch): validate them first
        if (kind != NS_KIND_UNALIGNED) {
            for (uint32_t s = 0; s < nseg && ok; ++s) { int64_t l; ok = seg_length(A.m, prm, key, s, epoch, l); }
        }
        int64_t remainder = 0; double ratio = 0;
        if (kind == NS_KIND_ALIGNED && ok) {                                         // S:1471-1474, 1351-1352
            uint32_t j = 0;
            for (; j < NS_KDE_RETRY; ++j) {
                u32x4 w = ns_draw(key, ST_HT, 0, a, j, 0);
                double x = ns_pow10m1(kde_sample(A.m.kde[NS_KDE_HT], w));
                if (x >= 0) { remainder = (int64_t)x; break; }
            }
            for (j = 0; j < NS_KDE_RETRY; ++j) {
                u32x4 w = ns_draw(key, ST_RATIO, 0, a, j, 0);
                double x = kde_sample(A.m.kde[NS_KDE_RATIO], w);
                if (0 <= x && x <= 1) { ratio = x; break; }
            }
            if (j == NS_KDE_RETRY) ratio = 0.5;
        }
        u32x4 ws = ns_draw(key, ST_STRAND, 0, a, 0, 0);
        const bool reversed = u32_to_p(ws.x) > A.m.strandness_rate;                  // S:1312, S:1524-1525
        if (!ok) { ++epoch; fails = 0; continue; }

        // ---- error lists (S:1355-1365, S:1501) ----
        EvSink sink; sink.ev = ev_base; sink.cap = ev_cap; sink.n = 0; sink.shift = 0; sink.last_ins_len = 0; sink.overflow = false;
        int64_t total = remainder;
        uint32_t evn = 0;
        for (uint32_t pi = 0; pi < n_pieces; ++pi) {
            const bool is_gap = (kind == NS_KIND_UNALIGNED) || (pi & 1);
            const uint32_t sid = is_gap ? NS_GAP_SEG + (pi >> 1) : (pi >> 1);
            int64_t mlen;
            if (kind == NS_KIND_UNALIGNED) mlen = unaligned_length(A.m, prm, key, a);
            else if (is_gap) mlen = gap_length(A.m, key, pi >> 1, epoch);
            else seg_length(A.m, prm, key, pi >> 1, epoch, mlen);
            sink.ev = ev_base + evn; sink.cap = ev_cap > evn ? ev_cap - evn : 0; sink.n = 0; sink.shift = 0;
            EList e;
            if (kind == NS_KIND_PERFECT) { e.l_new = e.middle_ref = mlen; }
            else if (is_gap) e = dev_unaligned_error_list(A.m, mlen, key, sid, a, sink);
            else e = dev_error_list(A.m, mlen, key, sid, a, sink);
            ns_piece p;
            p.ref_gpos = 0; p.ev_off = A.ev_off[r] + evn; p.chrom = 0; p.pos = 0;
            p.ref_len = (uint32_t)(e.middle_ref < 0 ? 0 : e.middle_ref);
            p.out_len = (uint32_t)((e.middle_ref < 0 ? 0 : e.middle_ref) + sink.shift);
            p.n_ev = sink.n; p.kind = is_gap ? 1u : 0u;
            pc[pi] = p;
            evn += sink.n;
            if (!is_gap) total += e.l_new;                                           // S:1362
            if (kind == NS_KIND_UNALIGNED) total = e.middle_ref;                     // S:1503
        }
        if (sink.overflow) { overflow = true; break; }
        if (total < prm.min_len || total > prm.max_len) {                            // S:1367-1368, S:1503-1504
            if (kind == NS_KIND_UNALIGNED) continue;
            if (++fails >= NS_EPOCH_FAILS) { ++epoch; fails = 0; }
            continue;
        }
        int64_t head = 0, tail = 0;                                                  // S:1377-1382
        if (kind == NS_KIND_ALIGNED && remainder != 0) {
            head = (int64_t)rint((double)remainder * ratio);
            tail = remainder - head;
        }
        // ---- positions (S:1388-1389, 1510, 1557) ----
        bool pos_ok = true;
        int64_t seq_len = head + tail;
        uint64_t ref_bases = 0;
        for (uint32_t pi = 0; pi < n_pieces; ++pi) {
            ns_piece p = pc[pi];
            const uint32_t sid = p.kind ? NS_GAP_SEG + (pi >> 1) : (pi >> 1);
            uint32_t chrom = 0; uint64_t pos = 0;
            if (p.kind && kind == NS_KIND_ALIGNED && gap_length(A.m, key, pi >> 1, epoch) == 0) {    // S:1553-1554
                p.ref_len = 0; p.out_len = 0; p.n_ev = 0;
            } else if (!extract_pos(A.ref, p.ref_len, key, sid, a, chrom, pos)) { pos_ok = false; break; }
            p.chrom = chrom; p.pos = (uint32_t)pos; p.ref_gpos = A.ref.chrom_off[chrom] + pos;
            pc[pi] = p;
            seq_len += p.out_len;
            ref_bases += p.ref_len;
        }
        if (!pos_ok) { ++epoch; fails = 0; continue; }
        if (seq_len < prm.min_len || seq_len > prm.max_len) { ++epoch; fails = 0; continue; }       // S:1429-1430, S:1518-1519

        // ---- accepted ----
        rd.reversed = reversed ? 1 : 0; rd.flags = 0;
        rd.head = (uint32_t)head; rd.tail = (uint32_t)tail; rd.seq_len = (uint32_t)seq_len; rd.attempts = a;
        // name length (S:1390-1402, 1332-1343, 1529-1534)
        uint32_t nl = 0; bool first = true;
        for (uint32_t pi = 0; pi < n_pieces; ++pi) {
            ns_piece p = pc[pi];
            if (p.kind && kind == NS_KIND_ALIGNED) continue;
            if (!first) nl += 2;            // ';' in the position list and ';' in the length list
            first = false;
            nl += (A.ref.name_off[p.chrom + 1] - A.ref.name_off[p.chrom] - 1) + 1 + dec_digits(p.pos) + dec_digits(p.ref_len);
        }
        nl += (kind == NS_KIND_ALIGNED ? 9u : kind == NS_KIND_PERFECT ? 9u : 11u) + dec_digits(prm.first_read + r);
        if (kind == NS_KIND_ALIGNED && nseg > 1) nl += 9;
        nl += 2 /*_F*/ + 1 + dec_digits((uint64_t)head) + 1 + 1 + dec_digits((uint64_t)tail);
        name_len = nl;
        if (prm.emit_errlog) {
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
        atomicAdd(&A.stats[1], (unsigned long long)seq_len);
        atomicAdd(&A.stats[2], (unsigned long long)ref_bases);
        atomicAdd(&A.stats[3], (unsigned long long)evn);
        done = true;
    }
    if (overflow) atomicAdd(&A.stats[0], 1ull);
    else if (!done) atomicAdd(&A.stats[4], 1ull);
    A.reads[r] = rd;
    A.name_len[r] = (uint16_t)name_len;
    uint64_t rl = 0;
    if (done && prm.emit_records)
        rl = (uint64_t)name_len + 2 + (uint64_t)rd.seq_len + 1 + (prm.fastq ? (uint64_t)rd.seq_len + 3 : 0);
    A.rec_len[r] = rl;
    A.err_len[r] = done ? err_len : 0;
}

// ---------------------------------------------------------------------------------------------------------
// k_names: record framing and read name, one thread per read
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_names(GenArgs A) {
    uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= A.prm.n_reads) return;
    ns_read rd = A.reads[r];
    rd.rec_off = A.rec_off[r];
    A.reads[r].rec_off = rd.rec_off;
    if (rd.flags || !A.prm.emit_records) return;
    const int kind = (int)A.prm.kind;
    const ns_piece *pc = A.pieces + rd.piece_off;
    uint8_t *p = A.records + rd.rec_off;
    *p++ = A.prm.fastq ? '@' : '>';
    bool first = true;
    for (uint32_t pi = 0; pi < rd.n_pieces; ++pi) {
        if (pc[pi].kind && kind == NS_KIND_ALIGNED) continue;
        if (!first) *p++ = ';';
        first = false;
        const char *cn = A.ref.names + A.ref.name_off[pc[pi].chrom];
        while (*cn) *p++ = (uint8_t)*cn++;
        *p++ = '_';
        p = put_dec(p, pc[pi].pos);
    }
    const char *tag = kind == NS_KIND_ALIGNED ? "_aligned_" : kind == NS_KIND_PERFECT ? "_perfect_" : "_unaligned_";
    while (*tag) *p++ = (uint8_t)*tag++;
    p = put_dec(p, A.prm.first_read + r);
    if (kind == NS_KIND_ALIGNED && rd.n_pieces > 1) { const char *c = "_chimeric"; while (*c) *p++ = (uint8_t)*c++; }
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
    *p++ = '_'; p = put_dec(p, rd.tail);
    *p++ = '\n';
    p += rd.seq_len;
    *p++ = '\n';
    if (A.prm.fastq) { *p++ = '+'; *p++ = '\n'; p += rd.seq_len; *p++ = '\n'; }
}

// ---------------------------------------------------------------------------------------------------------
// k_materialise: one read per wavefront (64-thread workgroup); see ns_materialise.h
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_materialise(GenArgs A, uint64_t nbases) {
    __shared__ TileLds T;
    const uint32_t lane = threadIdx.x;
    const uint64_t r = blockIdx.x;
    const ns_read rd = A.reads[r];
    if (rd.flags) return;
    const ns_key key = make_key(A.prm, r);
    const uint32_t a = rd.attempts;
    ReadOut ro;
    ro.seq = A.records + rd.rec_off + A.name_len[r] + 2;
    ro.qual = A.prm.fastq ? ro.seq + rd.seq_len + 3 : nullptr;
    ro.seq_len = rd.seq_len; ro.reversed = rd.reversed != 0;
    emit_random_region(A.m, ro, key, a, ST_HEAD, 0, rd.head, 0, lane);                                   // S:1426
    uint32_t q = rd.head;
    for (uint32_t pi = 0; pi < rd.n_pieces; ++pi) {
        const PieceCtx pc = load_piece(A.events, A.ref, A.pieces[rd.piece_off + pi], pi);
        materialise_piece(A.m, A.ref, T, ro, key, a, pc, q, lane, nbases);
        q += pc.out_len;
    }
    emit_random_region(A.m, ro, key, a, ST_TAIL, rd.seq_len - rd.tail, rd.tail, rd.head, lane);          // S:1427
}

// ---------------------------------------------------------------------------------------------------------
// k_errlog: error-profile rows "name\tpos\ttype\tlen\tref\tnew\n" in descending position order (S:1960, 2006-2008)
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_errlog(GenArgs A) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t r = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= A.prm.n_reads) return;
    const ns_read rd = A.reads[r];
    if (rd.flags) return;
    const ns_key key = make_key(A.prm, r);
    const uint32_t a = rd.attempts;
    const uint32_t nl = A.name_len[r];
    const uint8_t *name = A.records + rd.rec_off + 1;
    uint64_t base = A.err_off[r];
    for (uint32_t pi = 0; pi < rd.n_pieces; pi += 2) {
        const ns_piece p = A.pieces[rd.piece_off + pi];
        PieceCtx pc = load_piece(A.events, A.ref, p, pi);
        for (uint32_t j0 = 0; j0 < p.n_ev; j0 += 64) {
            // rows are written from the LAST event to the first
            const uint32_t k = j0 + lane;
            const bool active = k < p.n_ev;
            ns_event e; e.pos = 0; e.info = 0;
            if (active) e = pc.ev[p.n_ev - 1 - k];
            const uint32_t len = ns_ev_len(e.info), ty = ns_ev_type(e.info);
            uint32_t row = active ? nl + dec_digits(e.pos) + dec_digits(len) + 2u * len + 9u : 0;
            // exclusive prefix sum over the wavefront
            uint32_t incl = row;
            for (int off = 1; off < 64; off <<= 1) {
                uint32_t v = __shfl_up(incl, off);
                if ((int)lane >= off) incl += v;
            }
            const uint32_t total = __shfl(incl, 63);
            if (active) {
                uint8_t *q = A.errlog + base + (incl - row);
                for (uint32_t i = 0; i < nl; ++i) *q++ = name[i];
                *q++ = '\t'; q = put_dec(q, e.pos); *q++ = '\t';
                const char *tn = ty == NS_MIS ? "mis" : ty == NS_INS ? "ins" : "del";
                *q++ = (uint8_t)tn[0]; *q++ = (uint8_t)tn[1]; *q++ = (uint8_t)tn[2];
                *q++ = '\t'; q = put_dec(q, len); *q++ = '\t';
                uint8_t *q2 = q + len + 1;
                for (uint32_t i = 0; i < len; ++i) {
                    if (ty == NS_INS) { q[i] = '-'; q2[i] = ins_letter(key, pc.sid, a, e.pos, i); }
                    else {
                        uint32_t x = e.pos + i;
                        uint8_t cur = resolve_base(ref_base_at(A.ref, pc, x), key, pc.sid, a, x);
                        q[i] = cur;
                        q2[i] = (ty == NS_MIS) ? mis_letter(cur, key, pc.sid, a, e.pos, i) : (uint8_t)'-';
                    }
                }
                q[len] = '\t';
                q2[len] = '\n';
            }
            base += total;
        }
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
    hipStream_t stream = nullptr;
    std::string err;
    bool has_model = false, has_ref = false, has_batch = false;
    DevModel m{};
    DevRef ref{};
    std::vector<void *> model_allocs;
    void *ref_bases_owned = nullptr;
    std::vector<void *> ref_allocs;
    double cap_rate = 0.1;
    uint64_t ref_nbases = 0;
    // planning + result buffers
    DevBuf n_pieces, piece_off, ev_cap, ev_off, rec_len, rec_off, err_len, err_off, name_len;
    DevBuf reads, pieces, events, records, errlog, stats, scan_tmp;
    ns_batch_info last{};
    hipEvent_t evt[16]{};
    bool evt_ok = false;
};

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
    if (e != hipSuccess) { b.p = nullptr; b.cap = 0; return fail(ctx, NS_ENOMEM, std::string("hipMalloc: ") + hipGetErrorString(e)); }
    b.cap = want;
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

int ns_create(int device, ns_ctx **out) {
    if (!out) return NS_EINVAL;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return NS_ENODEV;
    if (device < 0 || device >= n) return NS_ENODEV;
    ns_ctx *ctx = new ns_ctx();
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return NS_EHIP;
    }
    for (auto &e : ctx->evt)
        if (hipEventCreate(&e) != hipSuccess) { delete ctx; return NS_EHIP; }
    ctx->evt_ok = true;
    *out = ctx;
    return NS_OK;
}

static void free_pool(std::vector<void *> &pool) {
    for (void *p : pool) { hipError_t e = hipFree(p); (void)e; }
    pool.clear();
}

void ns_destroy(ns_ctx *ctx) {
    if (!ctx) return;
    hipError_t e = hipSetDevice(ctx->device); (void)e;
    if (ctx->stream) { e = hipStreamSynchronize(ctx->stream); e = hipStreamDestroy(ctx->stream); }
    free_pool(ctx->model_allocs);
    free_pool(ctx->ref_allocs);
    if (ctx->ref_bases_owned) e = hipFree(ctx->ref_bases_owned);
    DevBuf *bufs[] = {&ctx->n_pieces, &ctx->piece_off, &ctx->ev_cap, &ctx->ev_off, &ctx->rec_len, &ctx->rec_off,
                      &ctx->err_len, &ctx->err_off, &ctx->name_len, &ctx->reads, &ctx->pieces, &ctx->events,
                      &ctx->records, &ctx->errlog, &ctx->stats, &ctx->scan_tmp};
    for (DevBuf *b : bufs)
        if (b->p) e = hipFree(b->p);
    if (ctx->evt_ok)
        for (auto &ev : ctx->evt) e = hipEventDestroy(ev);
    delete ctx;
}

const char *ns_last_error(const ns_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

static int set_ref_meta(ns_ctx *ctx, const uint64_t *chrom_off, uint32_t nchrom, const uint8_t *circular,
                        const char *names, uint64_t names_len) {
    if (!chrom_off || !nchrom || !circular || !names) return fail(ctx, NS_EINVAL, "reference metadata missing");
    free_pool(ctx->ref_allocs);
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

int ns_set_reference(ns_ctx *ctx, const uint8_t *bases, uint64_t nbases, const uint64_t *chrom_off, uint32_t nchrom,
                     const uint8_t *circular, const char *names, uint64_t names_len) {
    if (!ctx) return NS_EINVAL;
    if (!bases || !nbases) return fail(ctx, NS_EINVAL, "empty reference");
    HIPCHK(hipSetDevice(ctx->device));
    ctx->has_ref = false;
    if (ctx->ref_bases_owned) { HIPCHK(hipFree(ctx->ref_bases_owned)); ctx->ref_bases_owned = nullptr; }
    HIPCHK(hipMalloc(&ctx->ref_bases_owned, nbases + 16));
    HIPCHK(hipMemcpy(ctx->ref_bases_owned, bases, nbases, hipMemcpyHostToDevice));
    uint64_t nthreads = (nbases + 15) / 16;
    k_normalise<<<dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, ctx->stream>>>(
        static_cast<uint8_t *>(ctx->ref_bases_owned), nbases);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->ref.bases = static_cast<const uint8_t *>(ctx->ref_bases_owned);
    int rc = set_ref_meta(ctx, chrom_off, nchrom, circular, names, names_len);
    if (rc) return rc;
    if (chrom_off[nchrom] != nbases) return fail(ctx, NS_EINVAL, "chrom_off[nchrom] != nbases");
    ctx->ref_nbases = nbases;
    ctx->has_ref = true;
    return NS_OK;
}

int ns_set_reference_device(ns_ctx *ctx, const void *bases_dev, uint64_t nbases, const uint64_t *chrom_off,
                            uint32_t nchrom, const uint8_t *circular, const char *names, uint64_t names_len) {
    if (!ctx) return NS_EINVAL;
    if (!bases_dev || !nbases) return fail(ctx, NS_EINVAL, "empty reference");
    HIPCHK(hipSetDevice(ctx->device));
    ctx->has_ref = false;
    if (ctx->ref_bases_owned) { HIPCHK(hipFree(ctx->ref_bases_owned)); ctx->ref_bases_owned = nullptr; }
    // normalise in place: the caller's buffer becomes the upper-case IUPAC form
    uint64_t nthreads = (nbases + 15) / 16;
    k_normalise<<<dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, ctx->stream>>>(
        static_cast<uint8_t *>(const_cast<void *>(bases_dev)), nbases);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->ref.bases = static_cast<const uint8_t *>(bases_dev);
    int rc = set_ref_meta(ctx, chrom_off, nchrom, circular, names, names_len);
    if (rc) return rc;
    if (chrom_off[nchrom] != nbases) return fail(ctx, NS_EINVAL, "chrom_off[nchrom] != nbases");
    ctx->ref_nbases = nbases;
    ctx->has_ref = true;
    return NS_OK;
}

int ns_load_model(ns_ctx *ctx, const ns_model_tables *t) {
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
    if (t->flags & NS_MODEL_HAS_QUALS)
        if ((rc = upload(ctx, pool, &t->qual_thr[0][0], (size_t)NS_Q_COUNT * NS_QUAL_LEVELS, &m.qual_thr))) return rc;
    memcpy(m.hp, t->hp, sizeof m.hp);
    m.hp_mis_rate = t->hp_mis_rate;
    ctx->has_model = true;
    return NS_OK;
}

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

int ns_generate(ns_ctx *ctx, const ns_params *prm, ns_batch_info *info) {
    if (!ctx) return NS_EINVAL;
    if (!prm || !info) return fail(ctx, NS_EINVAL, "null params/info");
    if (!ctx->has_model || !ctx->has_ref) return fail(ctx, NS_ESTATE, "ns_generate before ns_load_model/ns_set_reference");
    if (prm->kind > NS_KIND_PERFECT) return fail(ctx, NS_EINVAL, "bad kind");
    if (prm->kind != NS_KIND_PERFECT && !(ctx->m.flags & NS_MODEL_HAS_ERRORS)) return fail(ctx, NS_EINVAL, "model has no error tables");
    if (prm->kind == NS_KIND_UNALIGNED && !prm->use_lognormal && !(ctx->m.flags & NS_MODEL_HAS_UNALIGNED))
        return fail(ctx, NS_EINVAL, "model has no unaligned-length KDE");
    if (prm->fastq && !(ctx->m.flags & NS_MODEL_HAS_QUALS)) return fail(ctx, NS_EINVAL, "model has no quality tables");
    if (prm->chimeric && !(ctx->m.flags & NS_MODEL_HAS_CHIMERIC)) return fail(ctx, NS_EINVAL, "model has no chimeric tables");
    if (prm->kmer_bias) return fail(ctx, NS_EINVAL, "homopolymer mode (-k) is not available in this build");
    if (prm->n_reads > 0x7ffffff0ull) return fail(ctx, NS_EINVAL, "batch too large (split into several calls)");
    if (prm->first_read + prm->n_reads >= (1ull << 40)) return fail(ctx, NS_EINVAL, "read index exceeds 2^40");
    HIPCHK(hipSetDevice(ctx->device));
    const size_t n = (size_t)prm->n_reads;
    memset(info, 0, sizeof *info);
    ctx->has_batch = false;
    if (!n) { ctx->last = *info; ctx->has_batch = true; return NS_OK; }
    int rc;
    if ((rc = ensure(ctx, ctx->n_pieces, (n + 1) * 4)) || (rc = ensure(ctx, ctx->piece_off, (n + 1) * 4)) ||
        (rc = ensure(ctx, ctx->ev_cap, (n + 1) * 8)) || (rc = ensure(ctx, ctx->ev_off, (n + 1) * 8)) ||
        (rc = ensure(ctx, ctx->rec_len, (n + 1) * 8)) || (rc = ensure(ctx, ctx->rec_off, (n + 1) * 8)) ||
        (rc = ensure(ctx, ctx->err_len, (n + 1) * 8)) || (rc = ensure(ctx, ctx->err_off, (n + 1) * 8)) ||
        (rc = ensure(ctx, ctx->name_len, (n + 1) * 2)) || (rc = ensure(ctx, ctx->reads, n * sizeof(ns_read))) ||
        (rc = ensure(ctx, ctx->stats, 8 * sizeof(unsigned long long))))
        return rc;

    GenArgs A;
    memset(&A, 0, sizeof A);
    A.prm = *prm; A.m = ctx->m; A.ref = ctx->ref;
    A.cap_gap_mul = 2;
    A.n_pieces = (uint32_t *)ctx->n_pieces.p; A.piece_off = (uint32_t *)ctx->piece_off.p;
    A.ev_cap = (uint64_t *)ctx->ev_cap.p; A.ev_off = (uint64_t *)ctx->ev_off.p;
    A.rec_len = (uint64_t *)ctx->rec_len.p; A.rec_off = (uint64_t *)ctx->rec_off.p;
    A.err_len = (uint64_t *)ctx->err_len.p; A.err_off = (uint64_t *)ctx->err_off.p;
    A.name_len = (uint16_t *)ctx->name_len.p; A.reads = (ns_read *)ctx->reads.p;
    A.stats = (unsigned long long *)ctx->stats.p;
    const dim3 blk(256);
    const dim3 grid_t((unsigned)((n + 1 + 255) / 256));        // thread-per-read kernels (n+1 for the scan sentinel)
    const dim3 grid_w((unsigned)((n + 3) / 4));                // wave-per-read kernels, 4 waves per block
    hipStream_t st = ctx->stream;
    unsigned long long stats[8];
    uint64_t tot_pieces = 0, tot_cap = 0;
    double cap_rate = ctx->cap_rate;
    HIPCHK(hipEventRecord(ctx->evt[0], st));
    for (int attempt = 0;; ++attempt) {
        A.cap_rate = cap_rate;
        HIPCHK(hipMemsetAsync(ctx->stats.p, 0, 8 * sizeof(unsigned long long), st));
        HIPCHK(hipEventRecord(ctx->evt[1], st));
        k_plan<<<grid_t, blk, 0, st>>>(A);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(ctx->evt[2], st));
        if ((rc = scan_u32(ctx, A.n_pieces, A.piece_off, n + 1))) return rc;
        if ((rc = scan_u64(ctx, A.ev_cap, A.ev_off, n + 1))) return rc;
        uint32_t tp32 = 0;
        HIPCHK(hipMemcpyAsync(&tp32, A.piece_off + n, 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(&tot_cap, A.ev_off + n, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        tot_pieces = tp32;
        if ((rc = ensure(ctx, ctx->pieces, (size_t)tot_pieces * sizeof(ns_piece) + 64)) ||
            (rc = ensure(ctx, ctx->events, (size_t)tot_cap * sizeof(ns_event) + 64)))
            return rc;
        A.pieces = (ns_piece *)ctx->pieces.p; A.events = (ns_event *)ctx->events.p;
        HIPCHK(hipEventRecord(ctx->evt[3], st));
        k_events<<<grid_t, blk, 0, st>>>(A);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(ctx->evt[4], st));
        if ((rc = scan_u64(ctx, A.rec_len, A.rec_off, n + 1))) return rc;
        if (prm->emit_errlog && (rc = scan_u64(ctx, A.err_len, A.err_off, n + 1))) return rc;
        HIPCHK(hipMemcpyAsync(stats, ctx->stats.p, sizeof stats, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(&info->record_bytes, A.rec_off + n, 8, hipMemcpyDeviceToHost, st));
        if (prm->emit_errlog) HIPCHK(hipMemcpyAsync(&info->errlog_bytes, A.err_off + n, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        info->n_overflow += stats[0];
        if (stats[0] == 0) break;
        if (attempt >= 6) return fail(ctx, NS_ENOMEM, "event capacity overflow persists after 6 retries");
        cap_rate *= 2.0; A.cap_gap_mul *= 2;          // rare: re-plan the batch with twice the event capacity
    }
    if (stats[4]) return fail(ctx, NS_EINVAL, "some reads found no acceptable length within the attempt limit "
                                              "(min_len/max_len too narrow for this model)");
    if ((rc = ensure(ctx, ctx->records, (size_t)info->record_bytes + 64)) ||
        (rc = ensure(ctx, ctx->errlog, (size_t)info->errlog_bytes + 64)))
        return rc;
    A.records = (uint8_t *)ctx->records.p; A.errlog = (uint8_t *)ctx->errlog.p;
    HIPCHK(hipEventRecord(ctx->evt[5], st));
    k_names<<<grid_t, blk, 0, st>>>(A);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(ctx->evt[6], st));
    if (prm->emit_records) {
        k_materialise<<<dim3((unsigned)n), dim3(64), 0, st>>>(A, ctx->ref_nbases);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(ctx->evt[7], st));
    if (prm->emit_errlog && prm->emit_records) {
        k_errlog<<<grid_w, blk, 0, st>>>(A);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(ctx->evt[8], st));
    HIPCHK(hipStreamSynchronize(st));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, ctx->evt[0], ctx->evt[8])); info->ms_total = ms;
    HIPCHK(hipEventElapsedTime(&ms, ctx->evt[1], ctx->evt[2])); info->ms_kernel[NS_K_LENGTHS] = ms;
    HIPCHK(hipEventElapsedTime(&ms, ctx->evt[3], ctx->evt[4])); info->ms_kernel[NS_K_EVENTS] = ms;
    HIPCHK(hipEventElapsedTime(&ms, ctx->evt[5], ctx->evt[6])); info->ms_kernel[NS_K_SCAN] = ms;   // names/framing
    HIPCHK(hipEventElapsedTime(&ms, ctx->evt[6], ctx->evt[7])); info->ms_kernel[NS_K_MATERIALISE] = ms;
    HIPCHK(hipEventElapsedTime(&ms, ctx->evt[7], ctx->evt[8])); info->ms_kernel[NS_K_ERRLOG] = ms;
    info->n_reads = n; info->n_pieces = tot_pieces; info->n_events = tot_cap;
    info->total_bases = stats[1]; info->total_ref_bases = stats[2]; info->events_used = stats[3];
    ctx->last = *info;
    ctx->last.n_events = tot_cap;
    ctx->has_batch = true;
    return NS_OK;
}

static int result_buf(ns_ctx *ctx, int which, const void **p, uint64_t *size) {
    const ns_batch_info &b = ctx->last;
    switch (which) {
        case NS_BUF_RECORDS: *p = ctx->records.p; *size = b.record_bytes; return NS_OK;
        case NS_BUF_READS: *p = ctx->reads.p; *size = b.n_reads * sizeof(ns_read); return NS_OK;
        case NS_BUF_PIECES: *p = ctx->pieces.p; *size = b.n_pieces * sizeof(ns_piece); return NS_OK;
        case NS_BUF_EVENTS: *p = ctx->events.p; *size = b.n_events * sizeof(ns_event); return NS_OK;
        case NS_BUF_ERRLOG: *p = ctx->errlog.p; *size = b.errlog_bytes; return NS_OK;
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

const void *ns_device_ptr(ns_ctx *ctx, int which) {
    if (!ctx || !ctx->has_batch) return nullptr;
    const void *p; uint64_t size;
    if (result_buf(ctx, which, &p, &size)) return nullptr;
    return p;
}

}  // extern "C"
