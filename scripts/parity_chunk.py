"""One chunk of scripts/parity_sweep.py in detail: python scripts/parity_chunk.py <mode index> <first read> — where GPU and oracle differ."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from nanosim_amd import engine as E, model as M  # noqa: E402
from tests import oracle_lib as O  # noqa: E402
import parity_sweep as PS  # noqa: E402

mi, first = int(sys.argv[1]), int(sys.argv[2])
name, rk, kw = PS.MODES[mi]
kw = dict(kw)
for item in os.environ.get("PC_SET", "").split(","):            # e.g. PC_SET=fastq=0,kmer_bias=0
    if item:
        k_, v_ = item.split("="); kw[k_] = type(kw.get(k_, 0))(int(v_))
mdl = M.load_model(os.path.join(PS.GOLDEN, "model_small", "training"), chimeric=True, homopolymer=True, fastq=True)
ref = M.read_fasta(os.path.join(PS.GOLDEN, "genome_small.fa" if rk == "lin" else "genome_circ.fa"), "linear" if rk == "lin" else "circular")
p = E.make_params(seed=PS.SEED0 + mi, first_read=first, n_reads=PS.CHUNK, max_len=kw.get("max_len", ref.max_chrom),
                  **{k: v for k, v in kw.items() if k != "max_len" and not k.startswith("_")})
eng = E.Engine(0); eng.set_reference(ref); eng.load_model(mdl)
start = int(sys.argv[3]) if len(sys.argv) > 3 else first
for f in range(start, first + 1, PS.CHUNK):
    p = E.make_params(seed=PS.SEED0 + mi, first_read=f, n_reads=PS.CHUNK, max_len=kw.get("max_len", ref.max_chrom),
                      **{k: v for k, v in kw.items() if k != "max_len" and not k.startswith("_")})
    b = eng.generate(p)
    got = dict(records=b.records(), errlog=b.errlog() if p.emit_errlog else np.zeros(0, np.uint8), reads=b.reads(), pieces=b.pieces(), events=b.events())
    exp = O.generate(mdl, ref, p, bytes_per_read=120000, events_per_read=24000)
    same = PS.digest(got) == PS.digest(exp)
    if not same or f == first:
        print("chunk", f, "identical" if same else "DIFFERS")
    if not same:
        break
rec, erec = got["records"], exp["records"]
print(name, "records equal:", rec.tobytes() == erec.tobytes(), len(rec), len(erec))
rd, er = got["reads"], exp["reads"]
for fld in ("n_pieces", "reversed", "flags", "head", "tail", "seq_len", "attempts"):
    if not np.array_equal(rd[fld], er[fld]):
        print("reads differ in", fld)
def gather(d):
    r = d["reads"]
    idx = np.concatenate([np.arange(int(o), int(o) + int(c)) for o, c in zip(r["piece_off"], r["n_pieces"])])
    return d["pieces"][idx]
pg, pe = gather(got), gather(exp)
print("pieces:", len(pg), len(pe))
for fld in ("ref_gpos", "pos", "ref_len", "out_len", "n_ev", "kind"):
    if not np.array_equal(pg[fld], pe[fld]):
        w = np.nonzero(pg[fld] != pe[fld])[0][:10]
        print("pieces differ in", fld, w, pg[fld][w], pe[fld][w])
if len(rec) == len(erec):
    d = np.nonzero(rec != erec)[0]
    print("differing bytes:", len(d), d[:20])
    if len(d):
        lo = int(d[0]); print("gpu:", bytes(rec[lo - 8:lo + 40])); print("exp:", bytes(erec[lo - 8:lo + 40]))
        print("run lengths of differing offsets:", np.diff(d)[:40])

eg, ee = got["events"], exp["events"]
for i in range(len(pg)):
    a = eg[int(pg["ev_off"][i]):int(pg["ev_off"][i]) + int(pg["n_ev"][i])]; c = ee[int(pe["ev_off"][i]):int(pe["ev_off"][i]) + int(pe["n_ev"][i])]
    if len(a) != len(c) or not (np.array_equal(a["pos"], c["pos"]) and np.array_equal(a["info"], c["info"])):
        print("events of piece", i, "differ", len(a), len(c)); break
if len(rec) == len(erec) and len(d):
    ro = rd["rec_off"].astype(np.int64)
    r = int(np.searchsorted(ro, d[0], side="right") - 1)
    end = int(ro[r + 1]) if r + 1 < len(ro) else len(rec)
    dd = d[(d >= ro[r]) & (d < end)] - ro[r]
    text = bytes(rec[ro[r]:end]); nl = text.index(b"\n")
    pcs = got["pieces"][int(rd["piece_off"][r]):int(rd["piece_off"][r]) + int(rd["n_pieces"][r])]
    sl = int(rd["seq_len"][r]); q0 = nl + 1 + sl + 3
    print("read", r, text[:nl], "seq_len", sl, "reversed", int(rd["reversed"][r]), "head", int(rd["head"][r]), "tail", int(rd["tail"][r]))
    print("pieces (kind, chrom, pos, ref_len, out_len, n_ev):", [(int(q["kind"]), int(q["chrom"]), int(q["pos"]), int(q["ref_len"]), int(q["out_len"]), int(q["n_ev"])) for q in pcs])
    print("differing quality positions (in line order):", (dd - q0)[:200], "count", len(dd), "other reads affected:", len(d) - len(dd))
    print("chrom lens", [int(x) for x in np.diff(ref.chrom_off)])
for fld in ("out_len",):
    w = np.nonzero(pg[fld] != pe[fld])[0]
    for i in w[:3]:
        print("piece", int(i), {k: int(pg[k][i]) for k in ("kind", "chrom", "pos", "ref_len", "out_len", "n_ev")}, "oracle out_len", int(pe["out_len"][i]), "n_ev", int(pe["n_ev"][i]),
              "wraps" if int(pg["pos"][i]) + int(pg["ref_len"][i]) > int(np.diff(ref.chrom_off)[int(pg["chrom"][i])]) else "")
        r = int(np.searchsorted(np.cumsum(rd["n_pieces"]), i, side="right"))
        print(" read", r, {k: int(rd[k][r]) for k in ("n_pieces", "reversed", "head", "tail", "seq_len", "attempts")}, "oracle seq_len", int(er["seq_len"][r]))
        o, n_ = int(rd["rec_off"][r]), int(rd["rec_off"][r + 1]) if r + 1 < len(rd) else len(rec)
        oo = int(er["rec_off"][r]); on_ = int(er["rec_off"][r + 1]) if r + 1 < len(er) else len(erec)
        a, c = bytes(rec[o:n_]).split(b"\n"), bytes(erec[oo:on_]).split(b"\n")
        print(" name", a[0][:120])
        sa, sc = a[1], c[1]
        k = next((j for j in range(min(len(sa), len(sc))) if sa[j] != sc[j]), None)
        print(" first differing base at", k, "gpu", sa[max(0, k - 30):k + 30], "exp", sc[max(0, k - 30):k + 30])
        ev = eg[int(pg["ev_off"][i]):int(pg["ev_off"][i]) + int(pg["n_ev"][i])]
        pos_ = ev["pos"].astype(np.int64); info = ev["info"].astype(np.uint32)
        ln = info & 0xfff; ty = (info >> 12) & 3; sh = (info >> 14).astype(np.int64) - 131072
        os_ = pos_ + sh
        sel = np.nonzero((os_ > k - int(rd["head"][r]) - 120) & (os_ < k - int(rd["head"][r]) + 60))[0]
        print(" events near (idx, pos, out_start, type, len):", [(int(j), int(pos_[j]), int(os_[j]), int(ty[j]), int(ln[j])) for j in sel])
        print(" n events with out_start < that:", int(np.sum(os_ < k - int(rd["head"][r]))))
