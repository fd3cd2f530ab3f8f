#!/usr/bin/env python3
"""Golden fixture of the training-side histogramming (second half): the REAL reference's
src/besthit_to_histogram.py:hist() is run in the build container on synthetic alignments and what it writes is committed as data.

hist(prefix, "bam") reads `<prefix>_primary.bam` through pysam, which this image lacks; the only thing it uses of an alignment is
`alnm.get_tag('cs')` (besthit_to_histogram.py:319-325), so the module is imported with a pysam stand-in whose AlignmentFile serves the
synthetic cs strings.  The cs strings are minimap2's short form (`:N` match, `*xy` mismatch, `+seq` insertion, `-seq` deletion), built
from the event lists of reads the CPU oracle generates with the small test model — so that they carry what real alignments carry:
errors next to each other (the mis0 / ins0 / del0 states), runs of mismatches, matches of every length.

    python tests/golden/make_hist_golden.py        -> tests/golden/reference_hist.json.gz
    python tests/golden/make_hist_golden.py --maf  -> tests/golden/reference_hist_maf.json.gz: hist(prefix, "maf") (B:188-315) on the two
                                                      `s` lines per alignment of a synthetic `<prefix>_besthit.maf`
"""
import gzip
import json
import os
import shutil
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF_SRC = "/root/reference/src"


def synthetic_cs(n_reads=450, seed=4242):
    """cs strings from oracle reads: the events of every aligned piece, walked in reference coordinates"""
    from nanosim_amd import engine as E
    from nanosim_amd import model as M
    from tests import oracle_lib as O
    golden = HERE
    mdl = M.load_model(os.path.join(golden, "model_small", "training"))
    ref = M.read_fasta(os.path.join(golden, "genome_small.fa"))
    p = E.make_params(seed=seed, first_read=0, n_reads=n_reads, max_len=ref.max_chrom)
    out = O.generate(mdl, ref, p)
    rng = np.random.default_rng(seed)
    letters = "acgt"
    cs = []
    for pc in out["pieces"]:
        if pc["kind"]:
            continue
        ev = out["events"][int(pc["ev_off"]):int(pc["ev_off"]) + int(pc["n_ev"])]
        s, pos = [], 0
        for e in ev:
            epos, ln, ty = int(e["pos"]), int(e["info"]) & 0xfff, (int(e["info"]) >> 12) & 3
            if epos > pos:
                s.append(":%d" % (epos - pos)); pos = epos
            seq = "".join(letters[i] for i in rng.integers(0, 4, ln))
            if ty == 0:                                   # mis: one *xy item per base
                s.append("".join("*" + a + letters[(letters.index(a) + 1 + int(rng.integers(0, 3))) % 4] for a in seq)); pos += ln
            elif ty == 1:
                s.append("+" + seq)
            else:
                s.append("-" + seq); pos += ln
        if int(pc["ref_len"]) > pos:
            s.append(":%d" % (int(pc["ref_len"]) - pos))
        cs.append("".join(s))
    # a few hand-made edge cases: one op only, errors at both ends, a long mismatch run, a match beyond 1000 (add_dict drops it, add_match
    # does not), junk between the items (re.findall skips it), an upper-case insertion
    cs += [":57", ":5*ag:3", ":1200*ct:900", ":3*ac*gt*ta*cg*ac*ag:7-tt+a:4", ":10+ACGT:5~gt12ag:6", ":8-a+c-g*at:2", ":4*aa", ":2+g:1-c:1*tg:1"]
    return cs


def run_reference(cs_list, workdir):
    class _Aln:
        def __init__(self, cs):
            self._cs = cs

        def get_tag(self, tag):
            if tag != "cs":
                raise KeyError(tag)
            return self._cs

    class _AlignmentFile:
        def __init__(self, path, mode):
            pass

        def fetch(self, until_eof=True):
            return iter(_Aln(c) for c in cs_list)
    pysam = types.ModuleType("pysam")
    pysam.AlignmentFile = _AlignmentFile
    sys.modules["pysam"] = pysam
    sys.dont_write_bytecode = True
    if REF_SRC not in sys.path:
        sys.path.insert(0, REF_SRC)
    import besthit_to_histogram as B
    prefix = os.path.join(workdir, "training")
    B.hist(prefix, "bam")
    files = {}
    for name in ("_match.hist", "_mis.hist", "_ins.hist", "_del.hist", "_error_markov_model", "_match_markov_model", "_first_match.hist",
                 "_error_rate.tsv"):
        files[name] = open(prefix + name).read()
    parsed = [list(map(list, B.parse_cs(c))) for c in cs_list[-8:]]          # parse_cs of the hand-made cases: (list_hist, list_op)
    cigar_md = [("10M", "10"), ("5M2I5M", "10"), ("4M1D6M", "4^A6"), ("3S7M", "3C3"), ("20M3I10M2D15M5S", "12A7T9^CG3G11"),
                ("2S8M1I4M1D9M", "3T4G3^A0C8"), ("50M", "0A48C0"), ("6M2D6M2I6M", "6^TT1A10")]
    # ... and 300 random alignments (soft clips at the ends; matches with mismatches inside, insertions, deletions in between; the MD
    # string derived the way aligners write it: match counts, mismatched reference bases, ^deleted bases, a 0 between two non-matches)
    rng = np.random.default_rng(99)
    for _ in range(300):
        ops, md, run = [], [], 0
        if rng.random() < 0.3:
            ops.append("%dS" % rng.integers(1, 30))
        n_blocks = int(rng.integers(1, 8))
        prev = None
        for b in range(n_blocks):
            kind = "M" if b % 2 == 0 or prev != "M" else ("I" if rng.random() < 0.5 else "D")
            if kind == "M":
                ln = int(rng.integers(1, 60))
                ops.append("%dM" % ln)
                i = 0
                while i < ln:
                    if rng.random() < 0.08:
                        md.append(str(run)); md.append("ACGT"[int(rng.integers(0, 4))]); run = 0
                    else:
                        run += 1
                    i += 1
            elif kind == "I":
                ops.append("%dI" % rng.integers(1, 6))
            else:
                ln = int(rng.integers(1, 6))
                ops.append("%dD" % ln)
                md.append(str(run)); md.append("^" + "".join("ACGT"[int(x)] for x in rng.integers(0, 4, ln))); run = 0
            prev = kind
        if prev != "M":
            ln = int(rng.integers(1, 40)); ops.append("%dM" % ln); run += ln
        md.append(str(run))
        if rng.random() < 0.3:
            ops.append("%dS" % rng.integers(1, 30))
        cigar_md.append(("".join(ops), "".join(md)))
    getcs = [[c, m, B.get_cs(c, m)] for c, m in cigar_md]                      # get_cs (B:76-132) by value
    return files, parsed, getcs


def synthetic_maf(n_reads=450, seed=777):
    """(reference line, query line) of pairwise alignments, as the two `s` lines of `<prefix>_besthit.maf` carry them (field 7), from oracle
    reads: the events of every aligned piece column by column, plus hand-made cases for the corners of the MAF branch (B:188-315)"""
    from nanosim_amd import engine as E
    from nanosim_amd import model as M
    from tests import oracle_lib as O
    mdl = M.load_model(os.path.join(HERE, "model_small", "training"))
    ref = M.read_fasta(os.path.join(HERE, "genome_small.fa"))
    p = E.make_params(seed=seed, first_read=0, n_reads=n_reads, max_len=ref.max_chrom)
    out = O.generate(mdl, ref, p)
    rng = np.random.default_rng(seed)
    L = "ACGT"
    pairs = []
    for pc in out["pieces"]:
        if pc["kind"]:
            continue
        ev = out["events"][int(pc["ev_off"]):int(pc["ev_off"]) + int(pc["n_ev"])]
        r, q, pos = [], [], 0

        def match(n):
            b = "".join(L[i] for i in rng.integers(0, 4, n))
            r.append(b); q.append(b)
        for e in ev:
            epos, ln, ty = int(e["pos"]), int(e["info"]) & 0xfff, (int(e["info"]) >> 12) & 3
            if epos > pos:
                match(epos - pos); pos = epos
            seq = [int(i) for i in rng.integers(0, 4, ln)]
            if ty == 0:
                r.append("".join(L[i] for i in seq)); q.append("".join(L[(i + 1 + int(rng.integers(0, 3))) % 4] for i in seq)); pos += ln
            elif ty == 1:
                r.append("-" * ln); q.append("".join(L[i] for i in seq))
            else:
                r.append("".join(L[i] for i in seq)); q.append("-" * ln); pos += ln
        if int(pc["ref_len"]) > pos:
            match(int(pc["ref_len"]) - pos)
        rs, qs = "".join(r), "".join(q)
        if rng.random() < 0.3:                                                 # soft-masked stretches: hist() upper-cases both lines
            a = int(rng.integers(0, max(1, len(rs) - 1))); b = min(len(rs), a + int(rng.integers(1, 200)))
            rs = rs[:a] + rs[a:b].lower() + rs[b:]
            if rng.random() < 0.5:
                qs = qs[:a] + qs[a:b].lower() + qs[b:]
        pairs.append([rs, qs])
    # corners: one match only; errors at both ends (pending counts at the end are dropped); a deletion directly in front of an insertion
    # and the other way round (both counters pending: the `elif` chain flushes ONE per column); mismatch next to indels (mis0 / ins0 /
    # del0); a match beyond 1 000 columns (add_dict drops it, add_match keeps it); lower against upper case; N against N
    pairs += [["ACGTACGT", "ACGTACGT"], ["A-CGT", "ATCGA"], ["ACG--T", "A-GTTT"], ["AC-GT", "ACT-T"], ["ACCGT--A", "A--GTTTA"],
              ["AAAA" + "C" * 1200 + "G-T", "AAAT" + "C" * 1200 + "GAT"], ["acgtNNac", "ACGTNNAC"], ["ACGT", "TGCA"], ["A-C", "AT-"],
              ["ACGTT-GCA-", "AC-TTAGCAT"], ["-ACGT", "TACGT"], ["ACGT-", "ACGTA"], ["TTGA--CCA", "TT--GGCCA"], ["AC--GT", "ACTT-T"]]
    return pairs


def run_reference_maf(pairs, workdir):
    for m in ("pysam",):
        sys.modules.setdefault(m, types.ModuleType(m))
    sys.dont_write_bytecode = True
    if REF_SRC not in sys.path:
        sys.path.insert(0, REF_SRC)
    import besthit_to_histogram as B
    prefix = os.path.join(workdir, "training")
    with open(prefix + "_besthit.maf", "w") as f:                              # two `s` lines per alignment, nothing else (B:190-198)
        for i, (r, q) in enumerate(pairs):
            f.write("s ref 0 %d + 1000000 %s\n" % (len(r.replace("-", "")), r))
            f.write("s read%d 0 %d + %d %s\n" % (i, len(q.replace("-", "")), len(q.replace("-", "")), q))
    B.hist(prefix, "maf")
    files = {}
    for name in ("_match.hist", "_mis.hist", "_ins.hist", "_del.hist", "_error_markov_model", "_match_markov_model", "_first_match.hist",
                 "_error_rate.tsv"):
        files[name] = open(prefix + name).read()
    return files


if __name__ == "__main__":
    if "--maf" in sys.argv:
        pairs = synthetic_maf()
        work = tempfile.mkdtemp(prefix="nshistmaf_")
        try:
            files = run_reference_maf(pairs, work)
        finally:
            shutil.rmtree(work, ignore_errors=True)
        out = os.path.join(HERE, "reference_hist_maf.json.gz")
        with gzip.open(out, "wt", compresslevel=9) as f:
            json.dump(dict(maf=pairs, files=files), f)
        print("written", out, os.path.getsize(out), "bytes;", len(pairs), "alignments;", {k: len(v) for k, v in files.items()})
        sys.exit(0)
    cs_list = synthetic_cs()
    work = tempfile.mkdtemp(prefix="nshist_")
    try:
        files, parsed, getcs = run_reference(cs_list, work)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    out = os.path.join(HERE, "reference_hist.json.gz")
    with gzip.open(out, "wt", compresslevel=9) as f:
        json.dump(dict(cs=cs_list, files=files, parse_cs_tail=parsed, get_cs=getcs), f)
    print("written", out, os.path.getsize(out), "bytes;", len(cs_list), "alignments;", {k: len(v) for k, v in files.items()})
