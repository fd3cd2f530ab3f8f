"""Large GPU == oracle sweep (run on the GPU box): every mode of the record path on many more reads than the -m gpu tests can afford
— the oracle runs on all host cores (fork pool, chunks of 500 reads), the engine generates the same index ranges; record images, error
profiles, per-read structs and event lists must be identical.
    python scripts/parity_sweep.py [reads_per_mode=40000]"""
import hashlib
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nanosim_amd import engine as E, model as M  # noqa: E402
from tests import oracle_lib as O  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
CHUNK = 500
SEED0 = int(os.environ.get("NS_SWEEP_SEED", str(0xC0FFEE)), 0)      # NS_SWEEP_SEED=...: other reads
MODES = [
    ("fasta+errlog", "lin", dict(emit_errlog=True)),
    ("fastq", "lin", dict(fastq=True)),
    ("fastq chimeric", "lin", dict(fastq=True, chimeric=True, emit_errlog=True)),
    ("unaligned fastq", "lin", dict(kind=E.NS_KIND_UNALIGNED, fastq=True)),
    ("unaligned long", "lin", dict(kind=E.NS_KIND_UNALIGNED, median_len=6000, sd_len=0.5)),
    ("fastq -k5", "lin", dict(fastq=True, kmer_bias=5, emit_errlog=True)),
    ("fasta -k4 chimeric", "lin", dict(kmer_bias=4, chimeric=True)),
    ("circular chimeric fastq", "circ", dict(fastq=True, chimeric=True)),
    ("circular unaligned", "circ", dict(kind=E.NS_KIND_UNALIGNED)),
    ("narrow window", "lin", dict(min_len=3000, max_len=9000, fastq=True)),
    ("circular chimeric fastq -k5", "circ", dict(fastq=True, chimeric=True, kmer_bias=5, emit_errlog=True)),
    ("perfect fastq", "circ", dict(kind=E.NS_KIND_PERFECT, fastq=True)),
    ("unaligned, background ctx", "lin", dict(kind=E.NS_KIND_UNALIGNED, fastq=True, median_len=3000, sd_len=0.7, _background=True)),
]
_CTX = {}


def digest(d):
    h = hashlib.sha256()
    for k in ("records", "errlog"):
        h.update(np.ascontiguousarray(d[k]).tobytes())
    r = d["reads"]
    for f in ("n_pieces", "reversed", "flags", "head", "tail", "seq_len", "attempts"):
        h.update(np.ascontiguousarray(r[f]).tobytes())
    # the pieces of every read through its piece_off (a chimeric read whose segment count changed with its epoch moved to fresh slots
    # behind the planned ones: the arrays of engine and oracle differ in their unused slots, not in what the reads point at)
    idx = np.concatenate([np.arange(int(o), int(o) + int(c)) for o, c in zip(r["piece_off"], r["n_pieces"])]) if len(r) else np.zeros(0, np.int64)
    p = d["pieces"][idx]
    for f in ("ref_gpos", "pos", "ref_len", "out_len", "n_ev", "kind"):
        h.update(np.ascontiguousarray(p[f]).tobytes())
    ev = d["events"]
    for i in range(len(p)):
        e = ev[int(p["ev_off"][i]):int(p["ev_off"][i]) + int(p["n_ev"][i])]
        h.update(np.ascontiguousarray(e["pos"]).tobytes()); h.update(np.ascontiguousarray(e["info"]).tobytes())
    return h.hexdigest()


def oracle_chunk(args):
    mode_i, first = args
    name, refk, kw = MODES[mode_i]
    mdl, refs = _CTX["mdl"], _CTX["refs"]
    ref = refs[refk]
    p = E.make_params(seed=SEED0 + mode_i, first_read=first, n_reads=CHUNK, max_len=kw.get("max_len", ref.max_chrom),
                      **{k: v for k, v in kw.items() if k != "max_len" and not k.startswith("_")})
    return digest(O.generate(mdl, ref, p, bytes_per_read=120000, events_per_read=24000))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
    mdl = M.load_model(os.path.join(GOLDEN, "model_small", "training"), chimeric=True, homopolymer=True, fastq=True)
    refs = {"lin": M.read_fasta(os.path.join(GOLDEN, "genome_small.fa"), "linear"),
            "circ": M.read_fasta(os.path.join(GOLDEN, "genome_circ.fa"), "circular")}
    _CTX.update(mdl=mdl, refs=refs)
    O.lib()
    jobs = [(mi, f) for mi in range(len(MODES)) for f in range(0, n, CHUNK)]
    with mp.get_context("fork").Pool(min(len(jobs), len(os.sched_getaffinity(0)))) as pool:
        res = pool.map_async(oracle_chunk, jobs, chunksize=1)
        # the engine meanwhile: same chunks (a read is a function of (seed, index), so chunking does not matter — checked by the tests)
        got = {}
        for refk in ("lin", "circ"):
            eng = E.Engine(0); eng.set_reference(refs[refk]); eng.load_model(mdl)
            os.environ["NS_COOP_MIN"] = "1"            # (read at ns_create: the background context routes small batches too)
            eng_bg = E.Engine(0); eng_bg.set_background(True); eng_bg.set_reference(refs[refk]); eng_bg.load_model(mdl)
            os.environ.pop("NS_COOP_MIN")
            fg = eng
            for mi, (name, rk, kw) in enumerate(MODES):
                if rk != refk:
                    continue
                eng = eng_bg if kw.get("_background") else fg
                for f in range(0, n, CHUNK):
                    p = E.make_params(seed=SEED0 + mi, first_read=f, n_reads=CHUNK, max_len=kw.get("max_len", refs[rk].max_chrom),
                                      **{k: v for k, v in kw.items() if k != "max_len" and not k.startswith("_")})
                    b = eng.generate(p)
                    got[(mi, f)] = digest(dict(records=b.records(), errlog=b.errlog() if p.emit_errlog else np.zeros(0, np.uint8),
                                               reads=b.reads(), pieces=b.pieces(), events=b.events()))
            fg.close(); eng_bg.close()
        exp = dict(zip(jobs, res.get()))
    bad = [k for k in jobs if got[k] != exp[k]]
    for mi, (name, _, _) in enumerate(MODES):
        nb = sum(1 for k in bad if k[0] == mi)
        print("%-26s %6d reads  %s" % (name, n, "identical" if not nb else "%d of %d chunks DIFFER (first at read %d)" % (nb, n // CHUNK, min(k[1] for k in bad if k[0] == mi))))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
