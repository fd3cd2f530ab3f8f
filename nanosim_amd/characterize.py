"""Training side, the histogramming step of the characterisation stage (second half): a drop-in for
``hist(prefix, "bam")`` of src/besthit_to_histogram.py (B:148-486) from the cs strings of the primary alignments on.

The reference walks every alignment in Python (parse_cs B:41-69, then the loop B:316-365) and fills dictionaries; the tables it writes
afterwards (B:366-486) are what ``simulator.py`` reads back as the error model (``read_profile``, src/simulator.py:473-501).  Here the
walk runs on the GPU through the C-ABI (``ns_cs_histograms``: one alignment per thread, include/nanosim_amd.h) and this module does
what is left: getting the cs strings out of a SAM file and formatting the reference's files from the counts, text for text.

    from nanosim_amd import characterize, engine
    eng = engine.Engine(0)
    characterize.hist("training", characterize.cs_from_sam("training_primary.sam"), eng)

    characterize.hist("training", characterize.maf_pairs("training_besthit.maf"), eng, alnm_ftype="maf")      # hist(prefix, "maf"), B:188-315

Not covered: BAM input (pysam is not a dependency here: convert with ``samtools view -h``).
"""
from __future__ import annotations

import ctypes as C
import re

import numpy as np

DICT_MAX = 1000                      # add_dict ignores larger values (B:15-16)
ERR_ROWS = ("mis", "ins", "del", "mis0", "ins0", "del0")
ERR_COLS = ("mis", "ins", "del")


class NsCsHist(C.Structure):
    """mirror of ns_cs_hist (include/nanosim_amd.h)"""
    _fields_ = [("cap_match2d", C.c_uint32), ("_pad", C.c_uint32), ("match_list", C.c_void_p), ("dic", (C.c_uint64 * 1001) * 5),
                ("error_list", C.c_uint64 * 18), ("first_error", C.c_uint64 * 3), ("max_match", C.c_uint64),
                ("n_match2d_overflow", C.c_uint64), ("n_skip", C.c_uint64), ("ms_kernel", C.c_double)]


_MD_TOKEN = re.compile(r'(\d+)|(\^[A-Za-z]+)|([A-Za-z])')
_CIGAR_TOKEN = re.compile(r'(\d+)([MIDSHX=])')          # (no N: introns are not looked for, as in the reference)


def get_cs(cigar_str: str, md_str: str) -> str:
    """The cs string the reference derives for an alignment that carries only CIGAR + MD (B:76-132).  It models indels and mismatches,
    not bases: a mismatch is always `*ab`, an inserted base `I`.  Written as a walk of the MD items over a cursor into the CIGAR; the
    reference's corner behaviour is kept because its histograms depend on it: an MD item that ends exactly where an M block ends closes
    the block as a MATCH of the remaining length (so a mismatch in the last column of a block counts as `:1`, and an item that starts at
    a block end leaves a `:0`), and the items behind an insertion start with whatever is left of the MD count."""
    blocks = [(int(n), op) for n, op in _CIGAR_TOKEN.findall(cigar_str)]
    out = []
    at = 0              # cursor into `blocks`
    block_end = 0       # query-side end of the blocks consumed so far (M, I and S advance it)
    md_pos = 0          # how far the MD items have come on the same axis
    for count, deleted, base in _MD_TOKEN.findall(md_str):
        if deleted:
            out.append("-" + deleted[1:])
            at += 1                                         # the D block that carries it
            continue
        left = int(count) if count else 1
        while at < len(blocks) and blocks[at][1] != 'D':
            size, op = blocks[at]
            if op == 'I':
                out.append("+" + "I" * size)
            elif op == 'M':
                if md_pos + left < block_end + size:        # the item ends inside this block
                    if left > 0:
                        out.append("*ab" if base else ":%d" % left)
                    md_pos += left
                    break
                rest = block_end + size - md_pos            # ... or runs to its end: the rest of the block is a match
                out.append(":%d" % rest)
                md_pos += rest
                left -= rest
                block_end += size
                at += 1
                continue
            elif op != 'S':                                 # (H, X, = : the reference never advances on these; minimap2 -a writes M/I/D/S)
                raise ValueError("CIGAR operation %r is not handled by get_cs (src/besthit_to_histogram.py:100-129)" % op)
            block_end += size
            md_pos += size
            at += 1
    return "".join(out)


def cs_from_sam(path: str):
    """the cs string of every alignment of a SAM text file: the cs:Z tag, else from CIGAR + MD:Z (B:320-324)"""
    out = []
    with open(path) as f:
        for line in f:
            if line.startswith("@"):
                continue
            fld = line.rstrip("\n").split("\t")
            if len(fld) < 11 or fld[5] == "*":
                continue
            cs = md = None
            for t in fld[11:]:
                if t.startswith("cs:Z:"):
                    cs = t[5:]
                elif t.startswith("MD:Z:"):
                    md = t[5:]
            if cs is None:
                if md is None:
                    raise ValueError("alignment %s has neither a cs nor an MD tag" % fld[0])
                cs = get_cs(fld[5], md)
            out.append(cs)
    return out


def count(eng, cs_list, cap: int = 2048) -> dict:
    """the counts of hist()'s loop for these alignments, from the GPU (ns_cs_histograms)"""
    blobs = [c.encode() if isinstance(c, str) else bytes(c) for c in cs_list]
    off = np.zeros(len(blobs) + 1, dtype=np.uint64)
    np.cumsum([len(b) for b in blobs], out=off[1:])
    data = np.frombuffer(b"".join(blobs) + b"\0", dtype=np.uint8)
    while True:
        h = NsCsHist()
        m2 = np.zeros((cap, cap), dtype=np.uint64)
        h.cap_match2d, h.match_list = cap, m2.ctypes.data
        eng._check(eng.L.ns_cs_histograms(eng.ctx, data.ctypes.data, int(off[-1]), off.ctypes.data, len(blobs), C.byref(h)))
        if h.n_skip:
            raise ValueError("long-form cs strings (`=` items) are not supported: the reference's parse_cs loses the pairing of its two "
                             "lists on them (src/besthit_to_histogram.py:49-65)")
        if not h.n_match2d_overflow:
            break
        cap = 1 << int(h.max_match).bit_length()                       # the matrix has to hold index max_match
    dic = np.ctypeslib.as_array(h.dic).copy()
    return dict(dic=dic, match_list=m2, error_list=np.ctypeslib.as_array(h.error_list).copy().reshape(6, 3),
                first_error=np.ctypeslib.as_array(h.first_error).copy(), max_match=int(h.max_match), ms_kernel=float(h.ms_kernel))


def maf_pairs(path: str):
    """[(reference line, query line)] of `<prefix>_besthit.maf` as hist(prefix, "maf") reads it (B:190-198): two `s` lines per alignment;
    field 7 of each is the aligned sequence (the upper-casing is done by the counting walk).  The file get_besthit_maf writes holds `s`
    lines only, and the reference assumes that; here everything else a MAF file may carry (`#` headers, `a score=` lines, blank
    separators) is skipped, and an `s` line without its partner is a ValueError."""
    out, pend = [], None
    with open(path) as f:
        for line in f:
            if not line.startswith("s ") and not line.startswith("s\t"):
                continue
            r = line.split()
            if pend is None:
                pend = r
                continue
            r, q, pend = pend, r, None
            if len(r) < 7 or len(q) < 7 or len(r[6]) > len(q[6]):
                raise ValueError("%s: not two `s` lines with an aligned sequence each (the reference would stop with an IndexError)" % path)
            out.append((r[6], q[6][:len(r[6])]))                   # (the walk runs over len(ref), B:203)
    if pend is not None:
        raise ValueError("%s: an `s` line without its partner (odd number of `s` lines)" % path)
    return out


def count_maf(eng, pairs, cap: int = 2048) -> dict:
    """the counts of hist()'s MAF loop for these alignments, from the GPU (ns_maf_histograms)"""
    rb = [a.encode() if isinstance(a, str) else bytes(a) for a, _ in pairs]
    qb = [b.encode() if isinstance(b, str) else bytes(b) for _, b in pairs]
    if any(len(a) != len(b) for a, b in zip(rb, qb)):
        raise ValueError("the two lines of an alignment differ in length")
    off = np.zeros(len(rb) + 1, dtype=np.uint64)
    np.cumsum([len(b) for b in rb], out=off[1:])
    ref = np.frombuffer(b"".join(rb) + b"\0", dtype=np.uint8)
    qry = np.frombuffer(b"".join(qb) + b"\0", dtype=np.uint8)
    while True:
        h = NsCsHist()
        m2 = np.zeros((cap, cap), dtype=np.uint64)
        h.cap_match2d, h.match_list = cap, m2.ctypes.data
        eng._check(eng.L.ns_maf_histograms(eng.ctx, ref.ctypes.data, qry.ctypes.data, int(off[-1]), off.ctypes.data, len(rb), C.byref(h)))
        if not h.n_match2d_overflow:
            break
        cap = 1 << int(h.max_match).bit_length()
    return dict(dic=np.ctypeslib.as_array(h.dic).copy(), match_list=m2, error_list=np.ctypeslib.as_array(h.error_list).copy().reshape(6, 3),
                first_error=np.ctypeslib.as_array(h.first_error).copy(), max_match=int(h.max_match), ms_kernel=float(h.ms_kernel))


def _dict_len(cnt, initial):
    nz = np.nonzero(cnt)[0]
    return max(initial, int(nz[-1]) + 1 if len(nz) else 0)


def format_tables(t: dict) -> dict:
    """{file suffix: text} exactly as hist() writes them (B:366-486) from the counts"""
    dic, m2 = t["dic"], t["match_list"]
    out = {}
    totals = {}
    for w, name, head, initial in ((0, "_match.hist", "Matches", 150), (2, "_mis.hist", "Mismatches", 30), (3, "_ins.hist", "Insertions", 30),
                                   (4, "_del.hist", "Deletions", 30)):
        n = _dict_len(dic[w], initial)
        lines = ["number of bases\t%s:\n" % head]
        tot = 0
        for key in range(n):
            lines.append(str(key) + "\t" + str(int(dic[w][key])) + "\n")
            tot += key * int(dic[w][key])
        out[name] = "".join(lines)
        totals[w] = tot
    total_match, total_mis, total_ins, total_del = totals[0], totals[2], totals[3], totals[4]
    den = total_mis + total_match + total_del
    out["_error_rate.tsv"] = ("Mismatch rate:\t" + str(total_mis * 1.0 / den) + '\n' + "Insertion rate:\t" + str(total_ins * 1.0 / den) + '\n' +
                              "Deletion rate:\t" + str(total_del * 1.0 / den) + '\n' +
                              "Total error rate:\t" + str((total_mis + total_ins + total_del) * 1.0 / den) + '\n')
    # error Markov model (B:404-422)
    err = t["error_list"]
    first = [int(x) for x in t["first_error"]]
    num_first = sum(first)
    s = "succedent \tmis\tins\tdel\n"
    s += "start\t" + str(first[0] * 1.0 / num_first) + "\t" + str(first[1] * 1.0 / num_first) + "\t" + str(first[2] * 1.0 / num_first)
    for r, x in enumerate(ERR_ROWS):
        s += "\n" + x
        pred = int(err[r].sum())
        for c in range(3):
            s += "\t" + ("0" if pred == 0 else str(int(err[r][c]) * 1.0 / pred))
    out["_error_markov_model"] = s
    # match Markov model (B:424-476).  The previous-match lengths are cut into <= 15 consecutive bins of about total / 15 pairs each:
    # a bin takes rows while it is below the target and stops in front of a row that would carry it further from the target than it is
    # (never in front of its first row); rows left over behind the 15th bin are added to its counts (its label keeps the old end)
    n = max(150, t["max_match"] + 1)
    pairs = np.zeros((n, n), dtype=np.int64)
    k = min(n, m2.shape[0])
    pairs[:k, :k] = m2[:k, :k]
    per_row = [int(x) for x in pairs.sum(axis=1)]
    target = sum(per_row) / 15
    edges, sizes = [], []                 # (first row, one past the last row) and the pairs of every bin
    row = 0
    while len(edges) < 15 and row < n:
        start, got = row, 0
        while got < target and row < n:
            if got != 0 and abs(got + per_row[row] - target) > abs(got - target):
                break
            got += per_row[row]
            row += 1
        edges.append((start, row))
        sizes.append(got)
    cols = [pairs[lo:hi].sum(axis=0) if hi > lo else np.zeros(n, dtype=np.int64) for lo, hi in edges]
    if row < n:
        cols[-1] = cols[-1] + pairs[row:n].sum(axis=0)
        sizes[-1] += sum(per_row[row:n])
    running = [0] * len(edges)
    lines = ["bins\t" + "\t".join("%s-%s" % e for e in edges) + '\n']
    for i in range(n):
        cells = [str(i) + "-" + str(i + 1)]
        for j in range(len(edges)):
            if sizes[j] == 0:
                cells.append("0")
            else:
                running[j] += int(cols[j][i]) * 1.0 / sizes[j]
                cells.append(str(running[j]))
        lines.append("\t".join(cells) + '\n')
    out["_match_markov_model"] = "".join(lines)
    # first match profile (B:478-486)
    nf = _dict_len(dic[1], 150)
    total_first = int(dic[1][:nf].sum())
    lines = ["bin\t0-50000\n"]
    cp = 0
    for i in range(nf):
        cp += int(dic[1][i]) * 1.0 / total_first
        lines.append(str(i) + "-" + str(i + 1) + "\t" + str(cp) + '\n')
    out["_first_match.hist"] = "".join(lines)
    return out


def hist(prefix: str, alignments, eng, alnm_ftype: str = "bam") -> dict:
    """writes <prefix>_match.hist, _mis.hist, _ins.hist, _del.hist, _error_rate.tsv, _error_markov_model, _match_markov_model and
    _first_match.hist like hist(prefix, alnm_ftype) (B:148-486; `prefix` may end in "_genome", B:150-151); returns the counts.
    alnm_ftype "bam" (or "sam"): alignments = cs strings (cs_from_sam); "maf": (reference line, query line) pairs (maf_pairs)"""
    if "_genome" in prefix:
        prefix = prefix[:-7]
    t = count_maf(eng, alignments) if alnm_ftype == "maf" else count(eng, alignments)
    for name, text in format_tables(t).items():
        with open(prefix + name, "w") as f:
            f.write(text)
    return t
