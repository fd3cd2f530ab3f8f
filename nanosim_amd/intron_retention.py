"""Intron-retention inputs of the transcriptome mode: ``<prefix>_IR_markov_model``, ``<prefix>_added_intron_final.gff3`` and the
genome FASTA (``-rg``).

Mirrors the ``model_ir`` half of ``read_profile`` (src/simulator.py:403-452) without HTSeq / pysam: the GFF3 lines are read with
the semantics of ``HTSeq.GFF_Reader(end_included=True)`` (0-based start = column 4 - 1, end = column 5, ``feature.name`` = value
of the FIRST attribute of column 9), the genome with the semantics of ``pysam.Fastafile`` (name = header up to the first white
space, bases as they are in the file).  ``eligible`` restates the transcript filter of the worker (S:1093-1099): with intron
retention switched on a transcript is only ever used when it is annotated and the lengths of its exons add up to its length.
"""
from __future__ import annotations

import ctypes as C
import re
from dataclasses import dataclass

import numpy as np

from .model import NsIrTables, Reference, read_fasta

NS_IR_EXON, NS_IR_INTRON = 0, 1
NS_IR_NO_CHROM = 0xFFFFFFFF
STATES = ("start", "no_IR", "IR")

_ATTR = re.compile(r"\s*([^\s=]+)[\s=]+(.*)")


def read_ir_markov_model(path: str):
    """-> (p_no_ir[3], p_ir[3]) for the rows start / no_IR / IR (S:414-422); the first line is a header"""
    rows = {}
    with open(path) as f:
        f.readline()
        for line in f:
            info = line.strip().split()
            if not info:
                continue
            rows[info[0]] = (float(info[1]), float(info[2]))
    missing = [s for s in STATES if s not in rows]
    if missing:
        raise SystemExit("IR Markov model %s lacks the row(s) %s" % (path, ", ".join(missing)))
    return [rows[s][0] for s in STATES], [rows[s][1] for s in STATES]


def _split_attributes(text: str):
    """';'-separated, quotes protect a ';' (HTSeq quotesafe_split)"""
    out, cur, quoted = [], [], False
    for ch in text:
        if ch == '"':
            quoted = not quoted
        if ch == ";" and not quoted:
            out.append("".join(cur))
            cur = []
        else:
            cur.append(ch)
    out.append("".join(cur))
    return out


def parse_gff_attributes(text: str):
    """-> (dict, value of the first attribute) as HTSeq.parse_GFF_attribute_string(text, True)"""
    d, first = {}, "_unnamed_"
    for i, a in enumerate(_split_attributes(text.rstrip("\n"))):
        if not a.strip():
            continue
        mo = _ATTR.match(a)
        if not mo:
            raise ValueError("Failure parsing GFF attribute line")
        val = mo.group(2)
        if val.startswith('"') and val.endswith('"'):
            val = val[1:-1]
        d[mo.group(1)] = val
        if i == 0:
            first = val
    return d, first


def read_structure(gff_path: str) -> dict:
    """dict_ref_structure (S:425-452): {transcript id: [(type, chrom, start, end, length, strand), ...]} in file order"""
    out: dict = {}
    with open(gff_path) as f:
        for line in f:
            if line == "\n" or line.startswith("#"):
                continue
            cols = line.split("\t", 8)
            if len(cols) < 9:
                raise ValueError("GFF3 line with fewer than 9 columns: " + line.strip())
            ftype = cols[2]
            if ftype != "exon" and ftype != "intron":
                continue
            attr, name = parse_gff_attributes(cols[8])
            if "transcript_id" in attr:
                fid = attr["transcript_id"]
            elif "Parent" in attr:
                info = name.split(":")
                if len(info) == 1:
                    fid = info[0]
                elif info[0] == "transcript":
                    fid = info[1]
                else:
                    continue
            else:
                continue
            fid = fid.split(".")[0]
            chrom = cols[0]
            if "chr" in chrom:
                chrom = chrom.strip("chr")                                     # S:448-449 (str.strip: a set of characters)
            start, end = int(cols[3]) - 1, int(cols[4])
            out.setdefault(fid, []).append((ftype, chrom, start, end, end - start, cols[6]))
    return out


def read_genome(path: str) -> Reference:
    """the genome FASTA as pysam.Fastafile serves it: raw names, bases in the case of the file"""
    return read_fasta(path, "linear", raw_names=True)


@dataclass
class IntronRetention:
    genome: Reference
    item_off: np.ndarray      # uint32 [n transcripts + 1]
    item_type: np.ndarray     # uint8
    item_minus: np.ndarray    # uint8
    item_chrom: np.ndarray    # uint32 (NS_IR_NO_CHROM: not in the genome FASTA)
    item_start: np.ndarray    # uint32
    item_len: np.ndarray      # uint32
    p_no_ir: list
    p_ir: list
    eligible: np.ndarray      # bool [n transcripts]

    def to_c(self) -> NsIrTables:
        t = NsIrTables()
        self._keep = [np.ascontiguousarray(self.genome.bases, dtype=np.uint8), np.ascontiguousarray(self.genome.chrom_off, dtype=np.uint64),
                      np.ascontiguousarray(self.item_off, dtype=np.uint32), np.ascontiguousarray(self.item_type, dtype=np.uint8),
                      np.ascontiguousarray(self.item_minus, dtype=np.uint8), np.ascontiguousarray(self.item_chrom, dtype=np.uint32),
                      np.ascontiguousarray(self.item_start, dtype=np.uint32), np.ascontiguousarray(self.item_len, dtype=np.uint32)]
        k = self._keep
        t.genome, t.genome_off = k[0].ctypes.data, k[1].ctypes.data
        t.n_gchrom, t.n_items = len(self.genome.names), len(self.item_type)
        t.item_off, t.item_type, t.item_minus = k[2].ctypes.data, k[3].ctypes.data, k[4].ctypes.data
        t.item_chrom, t.item_start, t.item_len = k[5].ctypes.data, k[6].ctypes.data, k[7].ctypes.data
        t.p_no_ir = (C.c_double * 3)(*self.p_no_ir)
        t.p_ir = (C.c_double * 3)(*self.p_ir)
        return t


def build(transcripts: Reference, structure: dict, genome: Reference, p_no_ir, p_ir) -> IntronRetention:
    """tables for ns_set_intron_retention: the structure of every transcript of the reference, chromosomes resolved against the genome"""
    flag_chrom = any("chr" in nm for nm in genome.names)                      # S:1066-1070
    gindex = {}
    for i, nm in enumerate(genome.names):
        gindex.setdefault(nm, i)
    glen = np.diff(genome.chrom_off.astype(np.int64))
    lens = np.diff(transcripts.chrom_off.astype(np.int64))
    index = {}
    for i, nm in enumerate(transcripts.names):
        index[nm] = i                                                       # a repeated id keeps its last record, as seq_dict does
    n = len(transcripts.names)
    per = [[] for _ in range(n)]
    for tid, items in structure.items():
        if tid in index:
            per[index[tid]] = items
    item_off = np.zeros(n + 1, dtype=np.uint32)
    ty, mi, ch, st, ln = [], [], [], [], []
    eligible = np.zeros(n, dtype=bool)
    for i in range(n):
        items = per[i]
        exon_len = 0
        for (ftype, chrom, start, end, length, strand) in items:
            name = ("chr" + chrom) if flag_chrom else chrom                   # S:1164-1166
            g = gindex.get(name, NS_IR_NO_CHROM)
            if g != NS_IR_NO_CHROM and (start < 0 or end > glen[g] or end < start):
                g = NS_IR_NO_CHROM                                            # (pysam would hand back a shorter string)
            if end < start:
                raise SystemExit("GFF3 feature with end < start in transcript " + transcripts.names[i])
            ty.append(NS_IR_INTRON if ftype == "intron" else NS_IR_EXON)
            mi.append(1 if strand == "-" else 0)
            ch.append(g); st.append(max(start, 0)); ln.append(length)
            if ftype == "exon":
                exon_len += length
        item_off[i + 1] = len(ty)
        eligible[i] = bool(items) and exon_len == int(lens[i])                # S:1095-1097
    return IntronRetention(genome=genome, item_off=item_off, item_type=np.array(ty, dtype=np.uint8), item_minus=np.array(mi, dtype=np.uint8),
                           item_chrom=np.array(ch, dtype=np.uint32), item_start=np.array(st, dtype=np.uint32),
                           item_len=np.array(ln, dtype=np.uint32), p_no_ir=list(p_no_ir), p_ir=list(p_ir), eligible=eligible)


def load(model_prefix: str, genome_path: str, transcripts: Reference) -> IntronRetention:
    p_no, p_ir = read_ir_markov_model(model_prefix + "_IR_markov_model")
    structure = read_structure(model_prefix + "_added_intron_final.gff3")
    return build(transcripts, structure, read_genome(genome_path), p_no, p_ir)
