// ns_ir.h — intron retention of transcriptome reads (src/simulator.py:114-191, 1156-1183).  Product code.
//
// The reference keeps, per transcript, the exon / intron items of the GFF3 annotation (dict_ref_structure, S:425-452) and a three
// state Markov chain over the introns (IR_markov_model, S:414-422).  Per aligned read, update_structure (S:114-145) walks the chain
// once (one uniform per intron); if at least one intron comes out as retained, extract_read_pos (S:148-191) cuts the read from the
// GENOME: exons and retained introns in GFF3 order, starting at most len_before bases into the transcript so that the first
// retained intron can be reached.  The intervals are recomputed from the Philox draws wherever they are needed (acceptance in
// k_chain, the read name in k_names, the copy in k_ir_splice) instead of being stored: (key, attempt) determine them.
#pragma once
#include "ns_device.h"

struct DevIr {
    const uint8_t *genome;            // bases of the genome FASTA as in the file (reverse_complement is case sensitive, S:1675-1680)
    const uint64_t *genome_off;       // [n_gchrom + 1]
    const uint32_t *item_off;         // [n transcripts + 1]
    const uint8_t *item_type;         // NS_IR_EXON / NS_IR_INTRON
    const uint8_t *item_minus;
    const uint32_t *item_chrom, *item_start, *item_len;
    double p_no_ir[3], p_ir[3];       // rows start / no_IR / IR
    uint8_t *arena;                   // spliced stretches of the batch (NS_BUF_SPLICED)
    const uint64_t *arena_off;        // [n reads + 1] exclusive scan of the slot sizes
};
enum : uint32_t { IR_ST_START = 0, IR_ST_NO = 1, IR_ST_YES = 2 };

// slot of a spliced stretch in the arena: NS_REF_PAD bytes in front (the copy loads up to 32 bytes before a segment), the bases,
// and padding up to a multiple of 16 with at least NS_REF_PAD bytes behind
#define NS_IR_PAD 64u
__host__ __device__ __forceinline__ uint32_t ir_slot_bytes(uint32_t ref_len) { return NS_IR_PAD + ((ref_len + NS_IR_PAD + 15u) & ~15u); }

// intron i of the transcript (S:123-133): p = random.random(); [0, p_no) -> no_IR, [p_no, p_no + p_ir) -> IR.  (A p beyond both
// intervals — the two probabilities of a row summing to less than 1 — appends nothing in the reference and later runs it out of
// list_states; it counts as no_IR here.)
__device__ __forceinline__ uint32_t ir_step(const DevIr &ir, uint32_t state, const ns_key &key, uint32_t a, uint32_t i) {
    const u32x4 w = ns_draw(key, ST_IR, 0, a, i >> 1, 0);
    const double p = (i & 1u) ? u53_to_p(w.z, w.w) : u53_to_p(w.x, w.y);
    if (p < ir.p_no_ir[state]) return IR_ST_NO;
    if (p < ir.p_no_ir[state] + ir.p_ir[state]) return IR_ST_YES;
    return IR_ST_NO;
}

struct IrPlan {
    bool any;                 // flag_ir of update_structure
    bool chrom_ok;            // every interval lies on a chromosome of the genome FASTA (S:1167-1169)
    bool minus;               // strand of the LAST interval (S:1177)
    uint32_t n_iv;
    uint32_t first_start;     // list_iv[0].start: the position in the read name (S:1175)
    uint32_t last_end;
    uint32_t struct_end;      // ref_trx_structure[-1][3]
    uint32_t name_extra;      // characters of "<start>-<end>;" over the retained intervals
};

// update_structure + extract_read_pos for a read of `length` reference bases of transcript `trx` (length trx_len);
// emit(interval index, genome chromosome, start, end, retained) for every genomic interval in GFF3 order
template <class F>
__device__ inline IrPlan ir_walk(const DevIr &ir, uint32_t trx, uint32_t length, uint32_t trx_len, const ns_key &key, uint32_t a, F &&emit) {
    IrPlan pl; pl.any = false; pl.chrom_ok = true; pl.minus = false; pl.n_iv = 0; pl.first_start = 0; pl.last_end = 0; pl.struct_end = 0; pl.name_extra = 0;
    const uint32_t i0 = ir.item_off[trx], i1 = ir.item_off[trx + 1];
    if (i1 == i0 || length == 0) return pl;
    uint32_t state = IR_ST_START, k = 0, len_before = 0;
    for (uint32_t i = i0; i < i1; ++i) {                                   // S:114-145, 153-159
        if (ir.item_type[i] == NS_IR_INTRON) { state = ir_step(ir, state, key, a, k++); if (state == IR_ST_YES) pl.any = true; }
        else if (!pl.any) len_before += ir.item_len[i];
    }
    if (!pl.any) return pl;
    const uint32_t hi = min(trx_len - length, len_before);                 // S:162: random.randint(0, min(ref_len - length, len_before))
    const u32x4 wp = ns_draw(key, ST_POS, 0, a, 0, 0);
    uint64_t sp64 = (uint64_t)(u53_to_p(wp.x, wp.y) * (double)((uint64_t)hi + 1));
    uint32_t start_pos = sp64 > hi ? hi : (uint32_t)sp64;
    uint32_t remaining = length;
    state = IR_ST_START; k = 0;
    for (uint32_t i = i0; i < i1 && remaining; ++i) {                      // S:164-184
        bool retained = false;
        if (ir.item_type[i] == NS_IR_INTRON) {
            state = ir_step(ir, state, key, a, k++);
            retained = state == IR_ST_YES;
            if (!retained) continue;
        }
        const uint32_t L = ir.item_len[i], s0 = ir.item_start[i];
        if (start_pos >= L) { start_pos -= L; continue; }
        const uint32_t start = s0 + start_pos;
        const uint32_t end = (L - start_pos >= remaining) ? start + remaining : s0 + L;
        remaining -= end - start;
        start_pos = 0;
        const uint32_t chrom = ir.item_chrom[i];
        emit(pl.n_iv, chrom, start, end, retained);
        if (pl.n_iv == 0) pl.first_start = start;
        pl.last_end = end; pl.minus = ir.item_minus[i] != 0;
        if (chrom == NS_IR_NO_CHROM) pl.chrom_ok = false;
        if (retained) pl.name_extra += dec_digits(start) + 1u + dec_digits(end) + 1u;
        ++pl.n_iv;
    }
    pl.struct_end = ir.item_start[i1 - 1] + ir.item_len[i1 - 1];
    if (pl.n_iv == 0) pl.any = false;
    return pl;
}
