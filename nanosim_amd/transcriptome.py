"""Transcriptome inputs: reference transcripts (-rt), expression profile (-e), polyA list (--polya).

Mirrors the transcriptome half of ``read_profile`` (src/simulator.py:341-350, 382-399, 460-470) and ``make_cdf`` (S:69-97).  The
transcripts become the "chromosomes" of one :class:`~nanosim_amd.model.Reference` (all linear); a transcript is picked with
``random.choices(ecdf_length_list, weights=ecdf_weight_list)`` (S:1084), i.e. by bisecting the running sum of the weights.
"""
from __future__ import annotations

import itertools
from dataclasses import dataclass

import numpy as np

from .model import Reference, read_fasta

POLYA_SCALE = {"albacore": 2.409858743694814}          # S:1046-1049; every other basecaller: 4.168299657168961
POLYA_SCALE_DEFAULT = 4.168299657168961


@dataclass
class TranscriptomeReference:
    ref: Reference
    expr_chrom: np.ndarray          # uint32 [n_expr]: transcript index, in the order of make_cdf (ascending expression)
    expr_cum: np.ndarray            # float64 [n_expr]: cumulative weights as random.choices accumulates them
    expr_weight: np.ndarray         # float64 [n_expr]: ecdf_weight_list
    polya: np.ndarray               # uint8 [n transcripts]: 1 = listed in --polya
    polya_scale: float


def read_expression(path: str) -> dict:
    """dict_exp: {transcript id without version: TPM} for TPM > 0, file order (S:382-396)"""
    out = {}
    with open(path) as f:
        f.readline()
        for line in f:
            parts = line.split("\t")
            if len(parts) < 3:
                raise SystemExit("Expression profile must contain 3 columns: ID, count, TPM ")
            tpm = float(parts[2])
            if tpm > 0:
                out[parts[0].split(".")[0]] = tpm
    if not out:
        raise SystemExit("Expression profile contains no TPM values > 0")
    return out


def make_cdf(dict_exp: dict, dict_len: dict):
    """S:69-97 -> (ecdf_length_list [(name, length)], ecdf_weight_list)"""
    sum_exp = 0
    matched = [k for k in dict_exp if k in dict_len]
    for k in matched:
        sum_exp += dict_exp[k]
    if not matched:
        raise SystemExit("Please make sure transcript IDs in the expression profile match with those in reference transcriptome "
                         "(example: both Ensembl IDs)")
    vals = [(k, dict_exp[k] / float(sum_exp)) for k in matched]
    vals.sort(key=lambda x: x[1])                                           # stable, as sorted() in the reference
    cdf = np.cumsum([v for _, v in vals])
    lo = np.concatenate([[0.0], cdf[:-1]])
    weights = np.abs(cdf - lo)
    return [(k, dict_len[k]) for k, _ in vals], [float(w) for w in weights]


def restrict_expression(tr: TranscriptomeReference, eligible: np.ndarray) -> TranscriptomeReference:
    """Intron retention on: the worker redraws the transcript until it is annotated with a consistent structure (S:1093-1099);
    drawing from the weights of the eligible transcripts alone is the same distribution."""
    keep = np.asarray(eligible, dtype=bool)[tr.expr_chrom]
    if not keep.any():
        raise SystemExit("No expressed transcript has a GFF3 structure whose exons add up to its length "
                         "(intron retention needs <prefix>_added_intron_final.gff3 to match the reference transcriptome)")
    w = tr.expr_weight[keep]
    return TranscriptomeReference(ref=tr.ref, expr_chrom=tr.expr_chrom[keep].copy(),
                                  expr_cum=np.array(list(itertools.accumulate(float(x) for x in w)), dtype=np.float64),
                                  expr_weight=w.copy(), polya=tr.polya, polya_scale=tr.polya_scale)


def read_transcriptome(fasta: str, expression: str, polya: str | None = None, basecaller: str | None = None) -> TranscriptomeReference:
    ref = read_fasta(fasta, "linear")
    lens = np.diff(ref.chrom_off.astype(np.int64))
    index = {}
    for i, nm in enumerate(ref.names):
        index[nm] = i                                                       # a repeated id keeps its last record, as the dict does
    dict_len = {nm: int(lens[i]) for nm, i in index.items()}
    names, weights = make_cdf(read_expression(expression), dict_len)
    flags = np.zeros(len(ref.names), dtype=np.uint8)
    if polya:
        with open(polya) as f:
            for line in f.readlines():
                tid = line.strip().split(".")[0]
                if tid in index:
                    flags[index[tid]] = 1
    return TranscriptomeReference(
        ref=ref, expr_chrom=np.array([index[k] for k, _ in names], dtype=np.uint32),
        expr_cum=np.array(list(itertools.accumulate(weights)), dtype=np.float64), expr_weight=np.array(weights, dtype=np.float64),
        polya=flags, polya_scale=POLYA_SCALE.get(basecaller or "", POLYA_SCALE_DEFAULT))
