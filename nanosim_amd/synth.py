"""Synthetic NanoSim model + reference generator.

The reference tree ships no pre-trained models (`.MISSING_LARGE_BLOBS`), so every
model used by tests and by bench.py is synthesised here in the *on-disk formats* the reference's
``read_profile`` parses (reference: src/simulator.py:244-591; writers cited per file below).

Nothing here is on the hot path; it only produces inputs.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field

import numpy as np

IUPAC_AMBIG = "YRWSKMDVHBN"


@dataclass
class SynthModelSpec:
    """Parameters of the `hg002_like` synthetic model."""
    n_train: int = 1_000_000
    seed: int = 20260926
    aligned_median: float = 7000.0
    aligned_sigma: float = 0.6
    ht_median: float = 40.0
    ht_sigma: float = 0.8
    ht_zero_frac: float = 0.02          # some reads have no head/tail at all
    unaligned_median: float = 1500.0
    unaligned_sigma: float = 0.9
    gap_median: float = 20.0
    gap_sigma: float = 1.0
    alignment_ratio: float = 19.0       # aligned/unaligned
    strandness: float = 0.5
    trx_median: float = 1400.0          # transcriptome: transcript lengths behind the 2-D KDE (_aligned_region_2d)
    trx_sigma: float = 0.6
    segment_mean: float = 1.05
    abun_inflation: float = 0.25        # metagenome only ("Shrinkage rate (beta)")
    # mixture parameters: [lambda, k, prob, weight]  (src/model_fitting.py:136,169,203)
    mis: tuple = (0.35, 0.0, 0.75, 0.55)
    ins: tuple = (1.1, 0.9, 0.55, 0.45)
    dele: tuple = (1.2, 0.95, 0.5, 0.5)
    # 7x3 error Markov model, rows start,mis,ins,del,mis0,ins0,del0 (src/besthit_to_histogram.py:410-422)
    trans: tuple = (
        (0.45, 0.22, 0.33),
        (0.46, 0.21, 0.33),
        (0.42, 0.25, 0.33),
        (0.44, 0.23, 0.33),
        (0.30, 0.30, 0.40),
        (0.50, 0.10, 0.40),
        (0.40, 0.35, 0.25),
    )
    ecdf_rows: int = 200
    # previous-match-length bins for the match Markov model and the mean match length in each
    mm_bins: tuple = ((0, 1), (1, 3), (3, 6), (6, 10), (10, 18), (18, 30), (30, 60), (60, 200))
    mm_means: tuple = (24.0, 26.0, 28.0, 30.0, 31.0, 32.0, 34.0, 36.0)
    mm_zero: tuple = (0.0, 0.02, 0.03, 0.03, 0.03, 0.03, 0.03, 0.03)   # P(match length 0)
    fm_mean: float = 18.0
    # base qualities: type -> (sd, loc, mu)   (src/model_base_qualities.py:82-96)
    quals: dict = field(default_factory=lambda: {
        "mis": (0.50, 0.0, 2.2), "ins": (0.48, 0.0, 2.3), "match": (0.35, 0.0, 3.2),
        "ht": (0.45, 0.0, 2.6), "unmapped": (0.42, 0.0, 2.4)})
    hp_mis_rate: float = 0.03
    # base -> dict of piecewise (const, beta1, breakpoint1, alpha1, alpha2) + lr (intercept, slope)
    hp: dict = field(default_factory=lambda: {
        "AT": dict(const=0.35, beta1=-0.25, breakpoint1=12.0, alpha1=0.92, alpha2=0.67,
                   intercept=0.0, slope=0.09),
        "CG": dict(const=0.42, beta1=-0.30, breakpoint1=10.0, alpha1=0.90, alpha2=0.60,
                   intercept=0.0, slope=0.10)})


def _geometric_cdf(n_rows: int, mean: float, p_zero: float, first_nonzero: int = 0) -> np.ndarray:
    """CDF over integer match lengths 0..n_rows-1 (row i is the bin i-(i+1))."""
    k = np.arange(n_rows, dtype=np.float64)
    q = 1.0 / max(mean, 1.0)
    pmf = q * (1.0 - q) ** k
    pmf[:first_nonzero] = 0.0
    if first_nonzero == 0:
        pmf[0] = 0.0
        pmf = pmf / pmf.sum() * (1.0 - p_zero)
        pmf[0] = p_zero
    else:
        pmf = pmf / pmf.sum()
    cdf = np.cumsum(pmf)
    cdf[-1] = 1.0
    return cdf


def write_model(prefix: str, spec: SynthModelSpec | None = None, *, write_pkl: bool = True,
                write_npz: bool = True) -> SynthModelSpec:
    """Write every `<prefix>_*` file of SURVEY.md §5.6.

    KDE training vectors are written both as sklearn/joblib pickles (what the reference loads,
    src/simulator.py:545-577; writers src/head_align_tail_dist.py:255-278) and as a neutral
    `<prefix>_kde.npz` that our loader prefers (no sklearn needed on the GPU box).
    """
    spec = spec or SynthModelSpec()
    rng = np.random.Generator(np.random.Philox(spec.seed))
    d = os.path.dirname(prefix)
    if d:
        os.makedirs(d, exist_ok=True)
    n = spec.n_train

    # --- KDE training vectors --------------------------------------------------------------
    aligned = np.maximum(1.0, np.rint(rng.lognormal(np.log(spec.aligned_median), spec.aligned_sigma, n)))
    ht_len = np.rint(rng.lognormal(np.log(spec.ht_median), spec.ht_sigma, n))
    ht_len[rng.random(n) < spec.ht_zero_frac] = 0.0
    ht = np.log10(ht_len + 1.0)
    ratio = rng.beta(2.0, 2.0, n)
    unaligned = np.maximum(1.0, np.rint(rng.lognormal(np.log(spec.unaligned_median), spec.unaligned_sigma, n)))
    gap = np.log10(np.rint(rng.lognormal(np.log(spec.gap_median), spec.gap_sigma, n)) + 1.0)
    # aligned_reads (perfect mode) = whole read length
    aligned_reads = aligned + ht_len
    # transcriptome: (transcript length, aligned length) pairs; most reads cover most of their transcript
    trx_len = np.maximum(200.0, np.rint(rng.lognormal(np.log(spec.trx_median), spec.trx_sigma, n)))
    trx_aligned = np.maximum(50.0, np.rint(trx_len * rng.beta(5.0, 1.5, n)))
    pairs = np.stack([trx_len, trx_aligned], axis=1)
    kdes = {
        "aligned_region": (aligned, 10.0), "aligned_reads": (aligned_reads, 10.0),
        "unaligned_length": (unaligned, 10.0), "ht_length": (ht, 0.01), "ht_ratio": (ratio, 0.01),
        "gap_length": (gap, 0.01),
    }
    if write_npz:
        np.savez(prefix + "_kde.npz", **{k + "_data": v[0] for k, v in kdes.items()},
                 **{k + "_bw": np.float64(v[1]) for k, v in kdes.items()},
                 aligned_region_2d_data=pairs, aligned_region_2d_bw=np.float64(10.0))
    if write_pkl:
        import joblib
        from sklearn.neighbors import KernelDensity
        for name, (vec, bw) in kdes.items():
            kde = KernelDensity(bandwidth=bw).fit(vec[:, None])
            joblib.dump(kde, prefix + "_" + name + ".pkl")
        joblib.dump(KernelDensity(bandwidth=10.0).fit(pairs), prefix + "_aligned_region_2d.pkl")

    # --- text tables -----------------------------------------------------------------------
    with open(prefix + "_model_profile", "w") as f:           # src/model_fitting.py:110,136,169,203
        f.write("Type\tlambda\tk\tprob\tweight\n")
        f.write("mismatch\t" + "\t".join(str(x) for x in spec.mis) + "\n")
        f.write("insertion\t" + "\t".join(str(x) for x in spec.ins) + "\n")
        f.write("deletion\t" + "\t".join(str(x) for x in spec.dele) + "\n")

    with open(prefix + "_error_markov_model", "w") as f:      # src/besthit_to_histogram.py:410-422
        f.write("succedent \tmis\tins\tdel\n")
        names = ["start", "mis", "ins", "del", "mis0", "ins0", "del0"]
        f.write("\n".join(nm + "\t" + "\t".join(str(x) for x in row) for nm, row in zip(names, spec.trans)))

    rows = spec.ecdf_rows
    with open(prefix + "_first_match.hist", "w") as f:        # src/besthit_to_histogram.py:479-484
        f.write("bin\t0-50000\n")
        cdf = _geometric_cdf(rows, spec.fm_mean, 0.0, first_nonzero=2)
        for i in range(rows):
            f.write("%d-%d\t%s\n" % (i, i + 1, str(float(cdf[i]))))

    with open(prefix + "_match_markov_model", "w") as f:      # src/besthit_to_histogram.py:464-474
        f.write("bins\t" + "\t".join("%d-%d" % b for b in spec.mm_bins) + "\n")
        cdfs = [_geometric_cdf(rows, m, z) for m, z in zip(spec.mm_means, spec.mm_zero)]
        for i in range(rows):
            f.write("%d-%d" % (i, i + 1))
            for c in cdfs:
                v = float(c[i])
                f.write("\t" + ("0" if v == 0.0 else str(v)))
            f.write("\n")

    with open(prefix + "_reads_alignment_rate", "w") as f:    # src/read_analysis.py:840-851
        f.write("Aligned / Unaligned ratio:\t" + str(spec.alignment_ratio) + "\n")
    with open(prefix + "_strandness_rate", "w") as f:         # src/read_analysis.py:833-835
        f.write("strandness:\t" + str(spec.strandness) + "\n")
    with open(prefix + "_chimeric_info", "w") as f:           # src/get_primary_sam.py:472-475
        f.write("Mean segments for each aligned read:\t" + str(spec.segment_mean) + "\n")
        f.write("Shrinkage rate (beta):\t" + str(spec.abun_inflation) + "\n")
    with open(prefix + "_base_qualities_model_parameters.tsv", "w") as f:   # src/model_base_qualities.py:82-96
        f.write("type\tsd\tloc\tmu\n")
        for t in ("mis", "ins", "match", "ht", "unmapped"):
            sd, loc, mu = spec.quals[t]
            f.write("%s\t%s\t%s\t%s\n" % (t, sd, loc, mu))
    with open(prefix + "_hp_lengths_model_parameters.tsv", "w") as f:       # src/model_homopolymer_lengths.py:236-243
        f.write("#Homopolymer mismatch rate: " + str(spec.hp_mis_rate) + "\n")
        cols = ["const", "beta1", "breakpoint1", "alpha1", "alpha2", "intercept", "slope"]
        f.write("base\t" + "\t".join(cols) + "\n")
        for base in ("AT", "CG"):
            f.write(base + "\t" + "\t".join(str(spec.hp[base][c]) for c in cols) + "\n")
    return spec


def synth_sequence(length: int, seed: int, *, n_frac: float = 0.0, iupac_frac: float = 0.0,
                   lower_frac: float = 0.0, hp_boost: float = 0.0) -> np.ndarray:
    """Random ASCII DNA (uint8). Optional N runs, sprinkled IUPAC codes, lower-case stretches and
    homopolymer enrichment so that `case_convert` (src/simulator.py:743-755) and `-k` have work."""
    rng = np.random.Generator(np.random.Philox(seed))
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    seq = lut[rng.integers(0, 4, length, dtype=np.uint8)]
    if hp_boost > 0 and length > 64:
        n_runs = int(length * hp_boost / 8)
        starts = rng.integers(0, length - 32, n_runs)
        lens = rng.geometric(0.25, n_runs) + 4
        for s, l in zip(starts, lens):
            seq[s:s + l] = seq[s]
    if n_frac > 0 and length > 1000:
        n_runs = max(1, int(length * n_frac / 200))
        starts = rng.integers(0, length - 400, n_runs)
        lens = rng.integers(50, 350, n_runs)
        for s, l in zip(starts, lens):
            seq[s:s + l] = ord("N")
    if iupac_frac > 0:
        k = int(length * iupac_frac)
        pos = rng.integers(0, length, k)
        codes = np.frombuffer(IUPAC_AMBIG.encode(), dtype=np.uint8)
        seq[pos] = codes[rng.integers(0, len(codes), k)]
    if lower_frac > 0 and length > 1000:
        n_runs = max(1, int(length * lower_frac / 300))
        starts = rng.integers(0, length - 600, n_runs)
        lens = rng.integers(100, 500, n_runs)
        for s, l in zip(starts, lens):
            seq[s:s + l] |= 0x20
    return seq


def write_fasta(path: str, records: list[tuple[str, np.ndarray]], width: int = 80) -> None:
    with open(path, "wb") as f:
        for name, seq in records:
            f.write(b">" + name.encode() + b"\n")
            b = seq.tobytes()
            for i in range(0, len(b), width):
                f.write(b[i:i + width] + b"\n")


# Reference sets of SURVEY.md §8d (lengths only; content is seeded random)
ECOLI_LEN = 4_641_652
CHR1_LEN = 248_956_422
GRCH38_LENS = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636,
               138394717, 133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345,
               83257441, 80373285, 58617616, 64444167, 46709983, 50818468, 156040895, 57227415]


def synth_transcriptome(n_trx: int, seed: int, *, median: float = 1400.0, sigma: float = 0.6, polya_frac: float = 0.6,
                        zero_tpm_frac: float = 0.2):
    """Synthetic transcriptome: [(name, bases)] with versioned Ensembl-like ids, an expression table (id, est_counts, tpm) in the
    layout of the -e file (S:382-399) and the ids of the transcripts carrying a polyA tail (--polya, S:460-470)."""
    rng = np.random.Generator(np.random.Philox(seed))
    lens = np.clip(np.rint(rng.lognormal(np.log(median), sigma, n_trx)), 150, 30000).astype(np.int64)
    recs = []
    for i, l in enumerate(lens):
        recs.append(("ENST%011d.%d" % (i + 1, 1 + i % 3), synth_sequence(int(l), seed * 1000003 + i, iupac_frac=0.0005 if i % 7 == 0 else 0.0)))
    tpm = rng.lognormal(1.0, 2.0, n_trx)
    tpm[rng.random(n_trx) < zero_tpm_frac] = 0.0
    polya = [recs[i][0] for i in range(n_trx) if rng.random() < polya_frac]
    expr = [(recs[i][0], float(tpm[i] * 3.0), float(tpm[i])) for i in range(n_trx)]
    return recs, expr, polya


def write_expression(path: str, expr) -> None:
    with open(path, "w") as f:
        f.write("target_id\test_counts\ttpm\n")
        for tid, cnt, tpm in expr:
            f.write("%s\t%s\t%s\n" % (tid, repr(cnt), repr(tpm)))


def _revcomp(seq: np.ndarray) -> np.ndarray:
    lut = np.arange(256, dtype=np.uint8)
    for a, b in (b"AT", b"TA", b"CG", b"GC", b"at", b"ta", b"cg", b"gc"):
        lut[a] = b
    return lut[seq[::-1]]


def synth_annotation(recs, seed: int, *, n_chrom: int = 3, chr_prefix: bool = True, max_exons: int = 6,
                     unannotated_frac: float = 0.1, p_model=((0.8, 0.2), (0.9, 0.1), (0.6, 0.4))):
    """Genome + GFF3 structure + IR Markov model for the transcripts `recs` of synth_transcriptome, in the layout the
    transcriptome mode reads with intron retention on (S:403-452): every annotated transcript is cut into exons, random introns
    (partly soft-masked, a few N) go between them, and the pre-mRNA is laid onto a strand of one of the chromosomes, so that the
    exons of a transcript add up to its sequence.  A few transcripts are left without structure, one gets exons that do not add
    up, one sits on a chromosome the genome FASTA lacks — the cases the worker filters (S:1095-1097, 1167-1169).
    -> (genome records, GFF3 text, Markov model text)"""
    rng = np.random.Generator(np.random.Philox(seed))
    chroms = [[] for _ in range(n_chrom)]           # list of arrays per chromosome
    clen = [0] * n_chrom
    lines = ["##gff-version 3"]

    def spacer(c, n):
        s = synth_sequence(n, int(rng.integers(1 << 30)), lower_frac=0.3)
        chroms[c].append(s); clen[c] += n

    for c in range(n_chrom):
        spacer(c, 500)
    for ti, (name, seq) in enumerate(recs):
        if rng.random() < unannotated_frac:
            continue
        L = len(seq)
        n_ex = int(min(max_exons, max(1, L // 120), 1 + rng.integers(0, max_exons)))
        cuts = np.sort(rng.choice(np.arange(1, L), size=n_ex - 1, replace=False)) if n_ex > 1 else np.array([], dtype=np.int64)
        bounds = [0] + [int(x) for x in cuts] + [L]
        exons = [seq[bounds[i]:bounds[i + 1]] for i in range(n_ex)]
        introns = []
        for _ in range(n_ex - 1):
            il = int(rng.integers(40, 600))
            s = synth_sequence(il, int(rng.integers(1 << 30)), lower_frac=0.5, iupac_frac=0.002)
            if rng.random() < 0.3:
                s[il // 3: il // 3 + 5] = ord("N")
            introns.append(s)
        parts, kinds = [], []
        for i in range(n_ex):
            parts.append(exons[i]); kinds.append("exon")
            if i + 1 < n_ex:
                parts.append(introns[i]); kinds.append("intron")
        minus = bool(rng.random() < 0.5)
        if minus:                                       # the genome carries the other strand; GFF3 order = ascending coordinates
            parts = [_revcomp(p) for p in parts[::-1]]
            kinds = kinds[::-1]
        c = int(rng.integers(0, n_chrom))
        missing_chrom = (ti % 37 == 5)
        cname = ("chr" if chr_prefix else "") + ("Un9" if missing_chrom else str(c + 1))
        style = ti % 4
        tid = name
        pos = clen[c]
        bad_sum = (ti % 41 == 7)
        lines.append("%s\tsynth\tgene\t%d\t%d\t.\t%s\t.\tID=gene:G%d" % (cname, pos + 1, pos + sum(len(p) for p in parts), "-" if minus else "+", ti))
        for k, (p, kind) in enumerate(zip(parts, kinds)):
            start1, end1 = pos + 1, pos + len(p)                       # GFF3: 1-based, end included
            if bad_sum and kind == "exon" and k == 0:
                end1 -= 1 if len(p) > 1 else 0
            if style == 0:
                attr = "transcript_id=%s;exon_number=%d" % (tid, k)
            elif style == 1:
                attr = "Parent=transcript:%s" % tid
            elif style == 2:
                attr = "Parent=%s;rank=%d" % (tid, k)
            else:
                attr = 'ID="%s:%s;%d";transcript_id=%s' % (kind, tid, k, tid)    # quoted value with a ';' inside
            lines.append("%s\tsynth\t%s\t%d\t%d\t.\t%s\t.\t%s" % (cname, kind, start1, end1, "-" if minus else "+", attr))
            chroms[c].append(p); clen[c] += len(p); pos += len(p)
        if ti % 5 == 0:                                # a feature the reader must skip: exon whose first attribute names no transcript
            lines.append("%s\tsynth\texon\t%d\t%d\t.\t+\t.\tID=exon:X%d;Parent=gene:G%d" % (cname, 1, 10, ti, ti))
        spacer(c, int(rng.integers(50, 400)))
    genome = [(("chr" if chr_prefix else "") + str(c + 1) + " synthetic chromosome", np.concatenate(chroms[c])) for c in range(n_chrom)]
    model = "state\tno_IR\tIR\n" + "".join("%s\t%r\t%r\n" % (st, p[0], p[1]) for st, p in zip(("start", "no_IR", "IR"), p_model))
    return genome, "\n".join(lines) + "\n", model


# zymo10_like: the ten species of the reference's sample_config_file/ (metagenome_list_for_simulation,
# dna_type_list.tsv: chromosome counts and topology; genome sizes of the ZymoBIOMICS standard) with the two abundance columns of
# abundance_for_simulation_multi_sample.tsv (even / log-distributed).  Content is seeded random sequence.
ZYMO10 = [  # (species, [(chromosome key, length, circular)], even abundance, log abundance)
    ("Bacillus_subtilis", [("BS-pilon-polished-v3-ST170922", 4_045_677, 1)], 12, 0.89),
    ("Cryptococcus_neoformans", [("NC-0267%d" % (45 + i), int(19_000_000 / 14 * (1.35 - 0.05 * i)), 0) for i in range(14)] + [("NC-018792", 24_919, 1)], 2, 0.00089),
    ("Enterococcus_faecalis", [("Enterococcus-faecalis-complete-genome", 2_845_392, 1)], 12, 0.00089),
    ("Escherichia_coli", [("Escherichia-coli-plasmid", 110_007, 1), ("Escherichia-coli-chromosome", 4_765_434, 1)], 12, 0.089),
    ("Lactobacillus_fermentum", [("Lactobacillus-fermentum-complete-genome", 1_905_333, 1)], 12, 0.0089),
    ("Listeria_monocytogenes", [("Listeria-monocytogenes-complete-genome", 2_992_342, 1)], 12, 89.1),
    ("Pseudomonas_aeruginosa", [("Pseudomonas-aeruginosa-complete-genome", 6_792_330, 1)], 12, 8.9),
    ("Saccharomyces_cerevisiae", [("NC-0011%d" % (33 + i), n, 0) for i, n in enumerate(
        [230_218, 813_184, 316_620, 1_531_933, 576_874, 270_161, 1_090_940, 562_643, 439_888, 745_751, 666_816, 1_078_177, 924_431, 784_333,
         1_091_291, 948_066])] + [("NC-001224", 85_779, 1)], 2, 0.89),
    ("Salmonella_enterica", [("Salmonella-enterica-plasmid1", 49_572, 1), ("Salmonella-enterica-chromosome", 4_759_746, 1)], 12, 0.089),
    ("Staphylococcus_aureus", [("Staphylococcus-aureus-chromosome", 2_718_780, 1), ("Staphylococcus-aureus-plasmid1", 6_339, 1),
                               ("Staphylococcus-aureus-plasmid2", 2_218, 1), ("Staphylococcus-aureus-plasmid3", 2_995, 1)], 12, 0.000089),
]


def zymo10_like(seed: int):
    """-> (chromosome names '<species>-<key>', concatenated bases, chrom_off, circular flags, species, species_chrom_off, keys per species,
    [even abundances, log abundances])"""
    names, chunks, circ, species, sp_off, keys = [], [], [], [], [0], []
    for si, (sp, chroms, _, _) in enumerate(ZYMO10):
        species.append(sp)
        keys.append([k for k, _, _ in chroms])
        for ci, (k, n, c) in enumerate(chroms):
            names.append(sp + "-" + k)
            chunks.append(synth_sequence(int(n), seed + 1000 * si + ci, n_frac=0.0005, iupac_frac=0.0002, lower_frac=0.02, hp_boost=0.005))
            circ.append(c)
        sp_off.append(sp_off[-1] + len(chroms))
    lens = np.array([len(c) for c in chunks], dtype=np.uint64)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    abun = [{sp: float(e) for sp, _, e, _ in ZYMO10}, {sp: float(g) for sp, _, _, g in ZYMO10}]
    return names, np.concatenate(chunks), off, np.array(circ, dtype=np.uint8), species, np.array(sp_off, dtype=np.uint32), keys, abun


def grch38_like(seed: int):
    """24 chromosomes with the GRCh38 primary lengths (3.1 Gb): (names, bases, chrom_off, circular)"""
    names = ["chr%d" % (i + 1) for i in range(22)] + ["chrX", "chrY"]
    chunks = [synth_sequence(n, seed + i, n_frac=0.0005, iupac_frac=0.0002, lower_frac=0.02, hp_boost=0.005) for i, n in enumerate(GRCH38_LENS)]
    off = np.concatenate([[0], np.cumsum(np.array(GRCH38_LENS, dtype=np.uint64))]).astype(np.uint64)
    return names, np.concatenate(chunks), off, np.zeros(24, dtype=np.uint8)
