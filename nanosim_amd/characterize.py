"""Training side, the histogramming step of the characterisation stage (second half): a drop-in for
``hist(prefix, "bam")`` of src/besthit_to_histogram.py (B:148-486) from the cs strings of the primary alignments on.

The reference walks every alignment in Python (parse_cs B:42-72, then the loop B:308-355) and fills dictionaries; the tables it writes
afterwards (B:357-486) are what ``simulator.py`` reads back as the error model (``read_profile``, src/simulator.py:473-501).  Here the
walk runs on the GPU through the C-ABI (``ns_cs_histograms``: one alignment per thread, include/nanosim_amd.h) and this module does
what is left: getting the cs strings out of a SAM file and formatting the reference's files from the counts, text for text.

    from nanosim_amd import characterize, engine
    eng = engine.Engine(0)
    characterize.hist("training", characterize.cs_from_sam("training_primary.sam"), eng)

Not covered: BAM input (pysam is not a dependency here: convert with ``samtools view -h``) and the MAF branch of the reference (B:187-306).
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


def get_cs(cigar_str: str, md_str: str) -> str:
    """the cs string of an alignment that carries only CIGAR + MD (B:79-130); arbitrary bases stand for the real ones, as there"""
    cs = []
    k = cx = cy = mx = my = 0
    md = re.findall(r'(\d+)|(\^[A-Za-z]+)|([A-Za-z])', md_str)
    cigar = re.findall(r'(\d+)([MIDSHX=])', cigar_str)
    for m in md:
        if m[1] != "":
            ln = len(m[1]) - 1
            cs.extend(["-", m[1][1:]])
            mx += ln; cx += ln; k += 1
        else:
            ml = int(m[0]) if m[0] != "" else 1
            while k < len(cigar) and cigar[k][1] != 'D':
                cl, op = int(cigar[k][0]), cigar[k][1]
                if op == "M":
                    if my + ml < cy + cl:
                        if ml > 0:
                            cs.extend(['*', 'a', 'b'] if m[2] != "" else [':', ml])
                        mx += ml; my += ml; ml = 0
                        break
                    dl = cy + cl - my
                    cs.extend([':', dl])
                    cx += cl; cy += cl; k += 1; mx += dl; my += dl; ml -= dl
                elif op == 'I':
                    cs.extend(['+', 'I' * cl])
                    cy += cl; my += cl; k += 1
                elif op == 'S':
                    cy += cl; my += cl; k += 1
                else:                                       # (H, X, = : the reference spins for ever on these; minimap2 -a emits M/I/D/S)
                    raise ValueError("CIGAR operation %r is not handled by get_cs (src/besthit_to_histogram.py:99-127)" % op)
    return "".join(str(x) for x in cs)


def cs_from_sam(path: str):
    """the cs string of every alignment of a SAM text file: the cs:Z tag, else from CIGAR + MD:Z (B:311-315)"""
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
                             "lists on them (src/besthit_to_histogram.py:50-66)")
        if not h.n_match2d_overflow:
            break
        cap = 1 << int(h.max_match).bit_length()                       # the matrix has to hold index max_match
    dic = np.ctypeslib.as_array(h.dic).copy()
    return dict(dic=dic, match_list=m2, error_list=np.ctypeslib.as_array(h.error_list).copy().reshape(6, 3),
                first_error=np.ctypeslib.as_array(h.first_error).copy(), max_match=int(h.max_match), ms_kernel=float(h.ms_kernel))


def _dict_len(cnt, initial):
    nz = np.nonzero(cnt)[0]
    return max(initial, int(nz[-1]) + 1 if len(nz) else 0)


def format_tables(t: dict) -> dict:
    """{file suffix: text} exactly as hist() writes them (B:357-486) from the counts"""
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
    # error Markov model (B:391-409)
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
    # match Markov model (B:411-466): 15 bins of the previous match length with about count / 15 events each
    n = max(150, t["max_match"] + 1)
    ml = np.zeros((n, n), dtype=np.int64)
    k = min(n, m2.shape[0])
    ml[:k, :k] = m2[:k, :k]
    row_sum = [int(x) for x in ml.sum(axis=1)]
    total = sum(row_sum)
    bin_size = total / 15
    k_of_bin = k_of_ml = last_k = 0
    count_each_bin, match_bin = {}, {}
    while k_of_bin < 15:
        if k_of_ml >= n:
            break
        match_bin[k_of_bin] = np.zeros(n, dtype=np.int64)
        tmp = 0
        while tmp < bin_size and k_of_ml < n:
            new_added = row_sum[k_of_ml]
            if abs(tmp + new_added - bin_size) > abs(tmp - bin_size) and tmp != 0:
                break
            tmp += new_added
            k_of_ml += 1
        if k_of_ml > last_k:
            match_bin[k_of_bin] += ml[last_k:k_of_ml].sum(axis=0)
        count_each_bin[k_of_bin] = [(last_k, k_of_ml), tmp]
        last_k = k_of_ml
        k_of_bin += 1
    if k_of_ml < n:
        match_bin[k_of_bin - 1] += ml[last_k:n].sum(axis=0)
        count_each_bin[k_of_bin - 1][1] += sum(row_sum[last_k:n])
    count_prob = [0] * len(match_bin)
    lines = ["bins\t" + "\t".join("%s-%s" % tup[0] for tup in count_each_bin.values()) + '\n']
    for i in range(n):
        row = [str(i) + "-" + str(i + 1)]
        for kb in match_bin:
            if count_each_bin[kb][1] == 0:
                row.append("\t" + "0")
            else:
                count_prob[kb] += int(match_bin[kb][i]) * 1.0 / count_each_bin[kb][1]
                row.append("\t" + str(count_prob[kb]))
        lines.append("".join(row) + '\n')
    out["_match_markov_model"] = "".join(lines)
    # first match profile (B:468-476)
    nf = _dict_len(dic[1], 150)
    total_first = int(dic[1][:nf].sum())
    lines = ["bin\t0-50000\n"]
    cp = 0
    for i in range(nf):
        cp += int(dic[1][i]) * 1.0 / total_first
        lines.append(str(i) + "-" + str(i + 1) + "\t" + str(cp) + '\n')
    out["_first_match.hist"] = "".join(lines)
    return out


def hist(prefix: str, cs_list, eng) -> dict:
    """writes <prefix>_match.hist, _mis.hist, _ins.hist, _del.hist, _error_rate.tsv, _error_markov_model, _match_markov_model and
    _first_match.hist like hist(prefix, "bam") (B:148-486; `prefix` may end in "_genome", B:150-151); returns the counts"""
    if "_genome" in prefix:
        prefix = prefix[:-7]
    t = count(eng, cs_list)
    for name, text in format_tables(t).items():
        with open(prefix + name, "w") as f:
            f.write(text)
    return t
