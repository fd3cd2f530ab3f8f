/*
 * nanosim_amd.h — C ABI of the MI355X-native read-generation engine.
 *
 * The reference (bcgsc/NanoSim v3.2.2, src/simulator.py) has NO FFI/plugin interface: its hot path is
 * a set of module-level Python functions that share ~30 globals filled by read_profile()
 * (src/simulator.py:244-591) and are entered through the mp.Process worker targets
 * (src/simulator.py:1601-1619, 1657-1660).  This header defines the boundary a maintainer would bind
 * at exactly that worker seam; each entry point cites what it replaces.
 *
 * Conventions: plain C, fixed-width little-endian integers, no exceptions across the ABI, every
 * function returns 0 on success or a negative NS_E* code (message via ns_last_error).  The CALLER owns
 * host buffers; the LIBRARY owns device buffers.  One context per GPU, used from one host thread.
 */
#ifndef NANOSIM_AMD_H
#define NANOSIM_AMD_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NS_ABI_VERSION 6u

/* error codes */
#define NS_OK 0
#define NS_EINVAL (-1)   /* bad argument / inconsistent tables */
#define NS_ENODEV (-2)   /* no HIP device / wrong architecture */
#define NS_ENOMEM (-3)   /* device or host allocation failed */
#define NS_EHIP (-4)     /* HIP runtime error (see ns_last_error) */
#define NS_ESTATE (-5)   /* call order violated (no model / no reference / no batch) */
#define NS_EIO (-6)      /* a file write of an output sink failed (ENOSPC, EBADF ...; see ns_last_error) */

/* ---- model tables: the flat form of the globals read_profile() fills (src/simulator.py:247-251) ---- */

/* One KDE (sklearn KernelDensity, gaussian): training vector + bandwidth.
 * Replaces kde.sample() at src/simulator.py:235 (i = floor(U*n); x = N(data[i], bw)). */
typedef struct ns_kde {
    const double *data;
    uint64_t n;
    double bw;
} ns_kde;

enum { NS_KDE_ALIGNED = 0,  /* _aligned_region.pkl (or _aligned_reads.pkl with --perfect), S:560,567 */
       NS_KDE_HT = 1,       /* _ht_length.pkl, log10(len+1), S:552 */
       NS_KDE_RATIO = 2,    /* _ht_ratio.pkl, S:555 */
       NS_KDE_UNALIGNED = 3,/* _unaligned_length.pkl, S:545 */
       NS_KDE_GAP = 4,      /* _gap_length.pkl, log10(len+1), S:577 */
       NS_KDE_COUNT = 5 };

/* error types / Markov states (trans_error_pr rows, src/simulator.py:486-495) */
enum { NS_MIS = 0, NS_INS = 1, NS_DEL = 2 };
enum { NS_ST_START = 0, NS_ST_MIS = 1, NS_ST_INS = 2, NS_ST_DEL = 3, NS_ST_MIS0 = 4, NS_ST_INS0 = 5, NS_ST_DEL0 = 6 };

/* quality classes (lognorm_base_qual keys, src/simulator.py:580-591) */
enum { NS_Q_MATCH = 0, NS_Q_MIS = 1, NS_Q_INS = 2, NS_Q_HT = 3, NS_Q_UNMAPPED = 4, NS_Q_COUNT = 5 };
#define NS_QUAL_LEVELS 128u
#define NS_HP_MAX_BREAKS 4u

typedef struct ns_hp_class {          /* one row (AT or CG) of _hp_lengths_model_parameters.tsv */
    double konst, alpha1;             /* predict_piecewise, src/model_homopolymer_lengths.py:167-186 */
    uint32_t n_breaks, _pad;
    double beta[NS_HP_MAX_BREAKS], breakpoint[NS_HP_MAX_BREAKS];
    double intercept, slope;          /* predict_lr, src/model_homopolymer_lengths.py:204-209 */
} ns_hp_class;

typedef struct ns_model_tables {
    uint32_t abi_version;             /* = NS_ABI_VERSION */
    uint32_t flags;                   /* NS_MODEL_* */

    /* ECDF of the first match length: read_ecdf(_first_match.hist), src/simulator.py:194-231,497-498.
     * Segment s covers (hi[s-1], hi[s]] (hi[-1] = 0) and maps linearly onto (vhi[s-1], vhi[s]),
     * vhi[-1] = fm_vlo0. */
    uint32_t fm_nseg, _pad0;
    const double *fm_hi, *fm_vhi;
    double fm_vlo0;

    /* match Markov model: read_ecdf(_match_markov_model), S:500-501; bins of previous match length */
    uint32_t mm_nbins, _pad1;
    const int64_t *mm_bin_lo, *mm_bin_hi;   /* [mm_nbins]  lo <= prev_match < hi, S:1891-1893 */
    const uint32_t *mm_seg_off;             /* [mm_nbins+1] offsets into mm_hi/mm_vhi */
    const double *mm_hi, *mm_vhi;
    const double *mm_vlo0;                  /* [mm_nbins] */

    /* error Markov model rows start,mis,ins,del,mis0,ins0,del0 (S:486-495):
     * trans[s][0] = a (mis is [0,a)), trans[s][1] = a+b (ins is [a,a+b)), trans[s][2] = 1-c (del is [1-c,1)) */
    double trans[7][3];

    /* run-length mixtures (error_par, S:473-484; samplers src/mixed_model.py:41-63) as inverse-CDF tables.
     * mix_cdf[t][0] = first component  (mis: Poisson(lambda)+1;  ins/del: ceil(lambda*Weibull(k)) with 0->1)
     * mix_cdf[t][1] = second component (mis: Geometric(p);       ins/del: Geometric(p)-1 with 0->1)
     * cdf[j] = P(value <= j+1); value = 1 + #{j : p > cdf[j]}, capped at mix_n. */
    double mix_w[3];
    uint32_t mix_n[3][2];
    const double *mix_cdf[3][2];

    ns_kde kde[NS_KDE_COUNT];

    double strandness_rate;           /* _strandness_rate or -s, S:270-275 */
    /* chimeric (S:571-577): number of segments ~ Geometric(1/segment_mean), table as above */
    uint32_t nseg_n, _pad2;
    const double *nseg_cdf;

    /* base qualities (src/model_base_qualities.py:9-20,120-130): per class, thr[j] = round(65536*P(q<=j));
     * q = #{j in [0,126] : h >= thr[j]} for a 16-bit draw h. */
    uint32_t qual_thr[NS_Q_COUNT][NS_QUAL_LEVELS];

    /* homopolymers (S:504-529; src/model_homopolymer_lengths.py:246-260) */
    ns_hp_class hp[2];                /* 0 = AT, 1 = CG */
    double hp_mis_rate;

    /* transcriptome: the 2-D KDE of (transcript length, aligned length) (_aligned_region_2d.pkl, S:561-565), training points sorted
     * by transcript length; select_nearest_kde2d (S:108-111) becomes a draw from the KDE conditioned on the transcript length */
    const double *kde2d_x, *kde2d_y;
    uint64_t kde2d_n;
    double kde2d_bw;
} ns_model_tables;

#define NS_MODEL_HAS_ERRORS 1u   /* error tables present (absent with --perfect) */
#define NS_MODEL_HAS_QUALS 2u
#define NS_MODEL_HAS_HP 4u
#define NS_MODEL_HAS_CHIMERIC 8u
#define NS_MODEL_HAS_UNALIGNED 16u
#define NS_MODEL_HAS_KDE2D 32u

/* ---- generation parameters: the arguments of the worker targets ------------------------------------
 * simulation_aligned_genome(dna_type, min_l, max_l, median_l, sd_l, out_reads, out_error, kmer_bias,
 *                           fastq, num_simulate, per, chimeric)            src/simulator.py:1266-1267
 * simulation_unaligned(dna_type, min_l, max_l, median_l, sd_l, out_reads, fastq, num_simulate, uracil)
 *                                                                          src/simulator.py:1482      */
enum { NS_KIND_ALIGNED = 0, NS_KIND_UNALIGNED = 1, NS_KIND_PERFECT = 2 };
#define NS_EMIT_SIZES 2u

typedef struct ns_params {
    uint64_t seed;          /* Philox key; (seed, read index) fully determine a read */
    uint64_t first_read;    /* global index of the first read of this batch (also the number in the name) */
    uint64_t n_reads;       /* num_simulate for this call */
    uint32_t kind;          /* NS_KIND_* */
    uint32_t fastq;         /* emit qualities / FASTQ records */
    uint32_t kmer_bias;     /* k of -k/--KmerBias; 0 = off (falsy in the reference, S:1413,1920) */
    uint32_t chimeric;
    uint32_t use_lognormal; /* -med/-sd given */
    uint32_t emit_records;  /* 1: format FASTA/FASTQ records on the device; NS_EMIT_SIZES: compute record_bytes / errlog_bytes of the batch
                             * without writing the images (sizing pass of a multi-rank run: every rank then writes at its final file offset) */
    int64_t min_len, max_len;
    double median_len, sd_len;
    uint32_t emit_errlog;   /* 1: format the _aligned_error_profile rows on the device (S:2006-2008); needs emit_records != 0 (the rows
                             * quote the read names of the record image): with emit_records = 0 errlog_bytes is 0 */
    uint32_t meta;          /* 1: metagenome batch = one worker of simulation_aligned_metagenome (S:814-1040) / simulation_unaligned("metagenome") */
    uint32_t trx;           /* 1: transcriptome batch = one worker of simulation_aligned_transcriptome (S:1043-1263) /
                             * simulation_unaligned("transcriptome"); needs ns_set_transcriptome */
    uint32_t uracil;        /* --uracil: T -> U in the emitted sequence (S:1247-1248) */
    uint32_t model_ir;      /* transcriptome, aligned reads: intron retention (S:1156-1183); needs ns_set_intron_retention */
    uint32_t reserved0;
} ns_params;

/* ---- device-side result layout (copied out with ns_copy_out) -------------------------------------- */

/* one e_dict entry (S:1875-1882), ascending order, position in un-mutated segment coordinates.
 * info = len[0:12] | type[12:14] | (shift + 2^17)[14:32]; shift = sum(ins - del) over the earlier events of
 * the same piece, so the payload of the event starts at emitted-segment offset pos + shift.  An attempt of a read with an event
 * outside these fields (multi-megabase reads only) is dropped and the read redrawn: ns_batch_info.n_range_redraws. */
typedef struct ns_event {
    uint32_t pos;           /* ceil(key): mis/del at pos, ins before index pos */
    uint32_t info;
} ns_event;
#define NS_EV_LEN_MAX 4095u
#define NS_EV_SHIFT_BIAS 131072
#define NS_EV_LEN(info) ((uint32_t)(info) & 0xfffu)
#define NS_EV_TYPE(info) (((uint32_t)(info) >> 12) & 3u)
#define NS_EV_SHIFT(info) ((int32_t)((uint32_t)(info) >> 14) - NS_EV_SHIFT_BIAS)
#define NS_EV_PACK(len, type, shift) (((uint32_t)(len) & 0xfffu) | ((uint32_t)(type) & 3u) << 12 | (uint32_t)((shift) + NS_EV_SHIFT_BIAS) << 14)

typedef struct ns_piece {   /* one aligned segment or one chimeric gap / unaligned body */
    uint64_t ref_gpos;      /* start offset in the concatenated reference; >= NS_SPLICED_BASE: offset of a spliced pre-mRNA
                             * stretch (intron retention) in the batch's splice arena (NS_BUF_SPLICED) */
    uint64_t ev_off;        /* first event in the event buffer */
    uint32_t chrom;
    uint32_t pos;           /* start inside the chromosome (the number in the read name, S:1747,1778) */
    uint32_t ref_len;       /* middle_ref: reference bases consumed (S:1363) */
    uint32_t out_len;       /* bases emitted for this piece */
    uint32_t n_ev;
    uint32_t kind;          /* 0 aligned segment, 1 gap / unaligned */
} ns_piece;

typedef struct ns_read {
    uint64_t rec_off;       /* byte offset of the record in the record buffer */
    uint32_t piece_off;     /* first piece */
    uint16_t n_pieces;      /* 2*segments-1 */
    uint8_t reversed;       /* is_reversed, S:1312 */
    uint8_t flags;
    uint32_t head, tail;    /* S:1377-1382 */
    uint32_t seq_len;       /* emitted bases incl. head/tail */
    uint32_t attempts;      /* rejected attempts before this one (S:1367) */
} ns_read;

typedef struct ns_batch_info {
    uint64_t n_reads, n_pieces;
    uint64_t n_events;      /* length of the event buffer (pieces index it through ev_off; capacity gaps included) */
    uint64_t events_used;   /* events actually generated */
    uint64_t record_bytes;  /* size of the FASTA/FASTQ image */
    uint64_t errlog_bytes;  /* size of the error-profile image */
    uint64_t total_bases;   /* sum of seq_len */
    uint64_t total_ref_bases; /* sum of ref_len over pieces (for the roofline byte count) */
    uint64_t n_overflow;    /* reads that needed the event-capacity fallback pass */
    double ms_total;        /* device time of the whole batch (HIP events on the engine stream) */
    double ms_kernel[8];    /* per-kernel device time: see NS_K_* */
    uint64_t spliced_bytes; /* intron retention: size of the splice arena (NS_BUF_SPLICED) */
    uint64_t n_range_redraws; /* attempts dropped because an event did not fit the 8-byte record (NS_EV_LEN_MAX / the shift field): the
                               * read drew new lengths, as after a failed final length check (S:1429-1430) */
} ns_batch_info;

enum { NS_K_LENGTHS = 0, NS_K_EVENTS = 1, NS_K_SCAN = 2, NS_K_MATERIALISE = 3, NS_K_HP = 4, NS_K_ERRLOG = 5,
       NS_K_RECORD_KERNEL = 6 };   /* 6: the record kernel alone (k_materialise; with -k: its last pass) — NS_K_MATERIALISE is the record STAGE:
                                    * that kernel, the generic kernel for the tiles it queued, the quality lines, the join with k_names */
enum { NS_BUF_RECORDS = 0, NS_BUF_READS = 1, NS_BUF_PIECES = 2, NS_BUF_EVENTS = 3, NS_BUF_ERRLOG = 4,
       NS_BUF_POLYA = 5 /* uint16 per read: polyA tail length of a transcriptome batch */,
       NS_BUF_SPLICED = 6 /* intron retention: the spliced stretches (transcript orientation, device form of the bases) */ };

typedef struct ns_ctx ns_ctx;

/* lifecycle */
int ns_create(int device, ns_ctx **out);
void ns_destroy(ns_ctx *ctx);
const char *ns_last_error(const ns_ctx *ctx);
uint32_t ns_abi_version(void);
/* A context whose worker calls run NEXT TO another context's on the same GPU (the reference runs its workers side by side, -t,
 * S:1588-1605; the unaligned worker call of a step runs like that).  A scheduling hint only — the reads are the same either way.
 * Until ABI 6 / round 5 it sent all but the longest eighth of a batch of unaligned reads through the thread-per-read error list;
 * since round 6 every unaligned read takes the wave-per-read one on either kind of context (NS_UCOOP_SHIFT=3 restores the split). */
int ns_set_background(ns_ctx *ctx, int on);

/* reference genome: replaces seq_dict/seq_len/genome_len (src/simulator.py:279-353).  `bases` is the
 * concatenation of all chromosomes as read from the FASTA (any case, IUPAC allowed); chrom_off has
 * nchrom+1 entries; circular[i] != 0 marks a circular chromosome (-dna_type circular, dict_dna_type);
 * names is a NUL-separated blob of the (already normalised, S:344-347) chromosome names. */
int ns_set_reference(ns_ctx *ctx, const uint8_t *bases, uint64_t nbases, const uint64_t *chrom_off,
                     uint32_t nchrom, const uint8_t *circular, const char *names, uint64_t names_len);
/* same, but `bases_dev` is already resident in this GPU's HBM (e.g. filled by an RCCL broadcast): the library makes a
 * device-to-device copy (normalised, padded for its unaligned loads); the caller's buffer is neither modified nor kept. */
int ns_set_reference_device(ns_ctx *ctx, const void *bases_dev, uint64_t nbases, const uint64_t *chrom_off,
                            uint32_t nchrom, const uint8_t *circular, const char *names, uint64_t names_len);

/* metagenome (src/simulator.py:284-339, 357-380): chromosomes of species s are [species_chrom_off[s], species_chrom_off[s+1]) of
 * the reference set with ns_set_reference (chromosome names are then "<species>-<chrom>", S:1747); abun = dict_abun of the
 * sample, abun_inflated = dict_abun_inflated (NULL unless chimeric, S:2511-2514).  ns_species_bases returns
 * current_species_bases (S:835, 1001-1002) of the last metagenome batch. */
int ns_set_species(ns_ctx *ctx, uint32_t nspecies, const uint32_t *species_chrom_off);
int ns_set_abundance(ns_ctx *ctx, const double *abun, const double *abun_inflated);
int ns_species_bases(ns_ctx *ctx, double *out);

/* transcriptome (src/simulator.py:341-350, 382-399, 460-470): the transcripts are the "chromosomes" of the reference set with
 * ns_set_reference (all linear).  expr_chrom / expr_cum: the transcripts with TPM > 0 in the order of make_cdf (S:69-97, ascending
 * expression) and the cumulative weights random.choices builds from ecdf_weight_list (S:1084); polya[nchrom] != 0 marks the
 * transcripts of the --polya list (NULL: none); polya_scale = scale of the exponential tail length (S:1046-1053). */
int ns_set_transcriptome(ns_ctx *ctx, uint32_t n_expr, const uint32_t *expr_chrom, const double *expr_cum,
                         const uint8_t *polya, double polya_scale);

/* intron retention (src/simulator.py:403-452: the GFF3 structure of the transcripts, the IR Markov model and the genome FASTA
 * that pysam serves in the reference).  Items of transcript t (a chromosome of the reference set with ns_set_reference) are
 * item_off[t] .. item_off[t+1]-1 in GFF3 order; coordinates are HTSeq's (0-based start, length = end - start).  A read of a
 * transcript in which update_structure (S:114-145) flags at least one intron as retained is cut from the genome by
 * extract_read_pos (S:148-191, 1159-1177): its piece has pos = the genome coordinate in the read name and
 * ref_gpos >= NS_SPLICED_BASE.  The host arrays are copied; NULL switches the feature off. */
typedef struct ns_ir_tables {
    const uint8_t *genome;          /* bases of the genome FASTA as they are in the file (case matters: S:1675-1680) */
    const uint64_t *genome_off;     /* [n_gchrom + 1] */
    uint32_t n_gchrom;
    uint32_t n_items;
    const uint32_t *item_off;       /* [n transcripts + 1] */
    const uint8_t *item_type;       /* NS_IR_EXON / NS_IR_INTRON */
    const uint8_t *item_minus;      /* 1: strand '-' */
    const uint32_t *item_chrom;     /* chromosome of the genome FASTA; NS_IR_NO_CHROM: not in the file (S:1167-1169) */
    const uint32_t *item_start;
    const uint32_t *item_len;
    double p_no_ir[3], p_ir[3];     /* rows start / no_IR / IR of <prefix>_IR_markov_model (S:414-422) */
} ns_ir_tables;
enum { NS_IR_EXON = 0, NS_IR_INTRON = 1 };
#define NS_IR_NO_CHROM 0xffffffffu
#define NS_SPLICED_BASE (1ull << 56)
int ns_set_intron_retention(ns_ctx *ctx, const ns_ir_tables *tables);

/* model: replaces the globals filled by read_profile() (src/simulator.py:473-591) */
int ns_load_model(ns_ctx *ctx, const ns_model_tables *tables);

/* the hot path: replaces one worker call simulation_aligned_genome / simulation_unaligned
 * (src/simulator.py:1266-1454, 1482-1549).  Results stay in HBM until the next ns_generate; the record and error-profile images
 * live in one of TWO result slots, so that a batch queued with ns_sink_write leaves the device while the next one is generated. */
int ns_generate(ns_ctx *ctx, const ns_params *params, ns_batch_info *info);

/* One STEP of simulation() (src/simulator.py:1571-1672): the aligned worker call (S:1601-1619 -> simulation_aligned_*) and the unaligned
 * one (S:1657-1660 -> simulation_unaligned) of the same share of the run, side by side on this GPU.  The reference joins the aligned
 * workers before it starts the unaligned ones (S:1621-1622); that order constrains the FILES only — the two calls write different files
 * and a read is a function of (seed, read index) — so the library runs them next to each other: the aligned call on `ctx`, the unaligned
 * one on the context's STEP COMPANION (own streams and batch buffers on the same device; it shares the reference, the model and the
 * mode tables of `ctx` — nothing is uploaded twice) from a worker thread the library keeps.  info[0] = the aligned batch, info[1] = the unaligned one; either params
 * pointer may be NULL (that call is skipped and its info zeroed).  The bytes of both batches are those of two ns_generate calls.
 * The unaligned batch's buffers belong to the companion: ns_step_context returns it (created on first use, owned and destroyed by
 * `ctx`; never pass it to ns_destroy) for ns_copy_out / ns_device_ptr / ns_record_offsets / ns_sink_* / ns_io_counters.
 * Added with ABI 6; ns_generate is unchanged. */
int ns_generate_step(ns_ctx *ctx, const ns_params *aligned, const ns_params *unaligned, ns_batch_info info[2]);
int ns_step_context(ns_ctx *ctx, ns_ctx **companion);

/* copy a result buffer of the last batch to host memory; nbytes must not exceed the buffer size
 * (record_bytes, n_reads*sizeof(ns_read), n_pieces*sizeof(ns_piece), n_events*sizeof(ns_event), errlog_bytes) */
int ns_copy_out(ns_ctx *ctx, int which, void *host_dst, uint64_t offset, uint64_t nbytes);
/* page-locked host memory for ns_copy_out destinations (the worker's out_reads / out_error writes, src/simulator.py:1437-1443,
 * 2006-2008, become DMA transfers at PCIe rate into such a buffer followed by plain file writes).  Owned by the caller. */
int ns_host_alloc(ns_ctx *ctx, uint64_t nbytes, void **out);
int ns_host_free(ns_ctx *ctx, void *p);

/* ---- output sinks: the worker's out_reads.write(...) / out_error.write(...) (src/simulator.py:1437-1443, 2006-2008) ----------------
 * A sink is an open file (descriptor owned by the caller) that the library appends result buffers to: ns_sink_write queues the record
 * image (NS_BUF_RECORDS) or the error-profile image (NS_BUF_ERRLOG) of the LAST batch and returns at once.  The bytes travel in
 * slices over the context's copy stream (DMA, concurrent with the kernels of the next ns_generate) into page-locked staging memory
 * and from there to the file with pwrite() at their final offsets, on the library's writer threads.  The next ns_generate fills the
 * other result slot; the one after it waits until the queued copies have left the device.  fd < 0: the bytes are copied to the host
 * and dropped.  ns_sink_put appends host bytes (the header line of the error profile, S:1634) in order with the queued buffers.
 * ns_sink_drain waits until everything queued is in the file and reports the first write error (NS_EIO); file_off (optional)
 * receives the offset behind the last byte.  Tuning (environment, read by the first ns_sink_open of a context): NS_IO_SLICE_MB
 * (16), NS_IO_SLICES (24), NS_IO_THREADS (16; at most one of them writes into a given file at a time). */
typedef struct ns_sink ns_sink;
int ns_sink_open(ns_ctx *ctx, int fd, uint64_t file_off, ns_sink **out);
int ns_sink_put(ns_ctx *ctx, ns_sink *sink, const void *host_src, uint64_t nbytes);
int ns_sink_write(ns_ctx *ctx, ns_sink *sink, int which);
/* the same for bytes [offset, offset + nbytes) of the buffer: one batch spread over several files (the reference's workers write
 * sub-files that are concatenated afterwards, S:1588-1639; several inodes are what lets the file writes run in parallel).
 * ns_record_offsets gives the cut points: for read indices of the last batch (0 .. n_reads; n_reads = the end) the byte offset of
 * the read's record in the record image and of its first row in the error-profile image (err_off may be NULL). */
int ns_sink_write_range(ns_ctx *ctx, ns_sink *sink, int which, uint64_t offset, uint64_t nbytes);
int ns_record_offsets(ns_ctx *ctx, const uint64_t *read_index, uint32_t n, uint64_t *rec_off, uint64_t *err_off);
int ns_sink_drain(ns_ctx *ctx, ns_sink *sink, uint64_t *file_off);
int ns_sink_close(ns_ctx *ctx, ns_sink *sink);
typedef struct ns_io_stats {
    uint64_t bytes;          /* bytes copied device -> host through the sinks of this context */
    double dma_ms;           /* sum of the slices' copy durations (HIP events on the copy stream): bytes / dma_ms = the DMA rate */
    double wait_staging_s;   /* time the copier waited for a free staging slice (the file writes are the limit when this is large) */
    double write_s;          /* sum of the writer threads' time in pwrite() */
    uint64_t slice_bytes;
    uint32_t n_slices, n_threads;
} ns_io_stats;
int ns_io_counters(ns_ctx *ctx, ns_io_stats *out, int reset);

/* ---- training side: the histograms of the characterisation stage ---------------------------------------------------------------------
 * replaces the counting loop of src/besthit_to_histogram.py:hist() (B:316-365 over parse_cs, B:41-69): from the cs strings of the primary
 * alignments (minimap2's short form: `:N` match, `*xy` mismatch, `+seq` insertion, `-seq` deletion) to
 *   dic[0..4]    run-length histograms of add_dict (B:14-22; values above 1000 are not counted): matches between errors, the first
 *                match of every alignment, mismatch runs, insertions, deletions           -> _match.hist, _first_match.hist, _mis/_ins/_del.hist
 *   match_list   (previous match, next match) counts of add_match (B:25-38; no upper limit) -> _match_markov_model
 *   error_list   error transitions, row = mis, ins, del, mis0, ins0, del0 (the previous error, "0": no match in between), column = mis,
 *                ins, del; first_error: the first error of every alignment                   -> _error_markov_model
 * cs: the strings back to back, aln_off[n_aln + 1] their offsets (host memory; copied to the device).  The tables the reference writes
 * from these counts are text formatting on the host (nanosim_amd/characterize.py).  match_list is a dense cap x cap matrix in caller-owned
 * host memory: an add_match with an index >= cap is counted in n_match2d_overflow and max_match says how large the matrix has to be. */
typedef struct ns_cs_hist {
    uint32_t cap_match2d;         /* in */
    uint32_t _pad;
    uint64_t *match_list;         /* in: host buffer of cap_match2d * cap_match2d counters (row = previous match length), or NULL */
    uint64_t dic[5][1001];        /* out: NS_CSH_MATCH, NS_CSH_FIRST_MATCH, NS_CSH_MIS, NS_CSH_INS, NS_CSH_DEL */
    uint64_t error_list[18];
    uint64_t first_error[3];
    uint64_t max_match;           /* largest length handed to add_match */
    uint64_t n_match2d_overflow;
    uint64_t n_skip;              /* `=` items (long-form cs): the reference's two lists fall out of step on them — refuse the input */
    double ms_kernel;
} ns_cs_hist;
enum { NS_CSH_MATCH = 0, NS_CSH_FIRST_MATCH = 1, NS_CSH_MIS = 2, NS_CSH_INS = 3, NS_CSH_DEL = 4 };
int ns_cs_histograms(ns_ctx *ctx, const uint8_t *cs, uint64_t nbytes, const uint64_t *aln_off, uint32_t n_aln, ns_cs_hist *h);
/* The MAF branch of the same loop (B:188-315): <prefix>_besthit.maf carries two `s` lines per alignment, the aligned reference and query
 * sequences with '-' for gaps.  ref_lines / query_lines: those lines (field 7) of all alignments back to back, alignment a at
 * aln_off[a] .. aln_off[a + 1] of BOTH (the two lines of an alignment have the same length; nbytes = aln_off[n_aln]).  Same counters,
 * same struct; per alignment the state of the reference's column walk (its four pending run lengths, prev_match and prev_error reset
 * per alignment).  Added with ABI 6. */
int ns_maf_histograms(ns_ctx *ctx, const uint8_t *ref_lines, const uint8_t *query_lines, uint64_t nbytes, const uint64_t *aln_off,
                      uint32_t n_aln, ns_cs_hist *h);

/* device address of a result buffer (for zero-copy consumers such as torch / RCCL); NULL if absent */
const void *ns_device_ptr(ns_ctx *ctx, int which);

#ifdef __cplusplus
}
#endif
#endif /* NANOSIM_AMD_H */
