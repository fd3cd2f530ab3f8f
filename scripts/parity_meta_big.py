"""Metagenome worker calls far larger than the -m gpu cases (tests/test_gpu_metagenome.py), GPU == oracle bit for bit:
    python scripts/parity_meta_big.py [reads per case = 20000]      (GPU box; the oracle is sequential: ~10 s per case)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
from nanosim_amd import engine as E, metagenome as MG, model as M  # noqa: E402
from tests import oracle_lib as O  # noqa: E402
from tests.test_gpu_parity import compare  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
META = os.path.join(ROOT, "tests", "golden", "meta")
mdl = M.load_model(os.path.join(ROOT, "tests", "golden", "model_small", "training"), chimeric=True, homopolymer=True, fastq=True)
ref = MG.read_metagenome(os.path.join(META, "genome_list.tsv"), os.path.join(META, "dna_type_list.tsv"))
_, samples = MG.read_abundance(os.path.join(META, "abundance.tsv"), ref.species)
abun = samples[0]
infl = {sp: MG.inflate_abun(abun, sp, mdl.abun_inflation) for sp in abun}
eng = E.Engine(0); eng.set_metagenome(ref, abun, infl); eng.load_model(mdl)
bad = 0
for name, kw in (("fastq chimeric errlog", dict(chimeric=True, fastq=True, emit_errlog=True)),
                 ("fastq -k5 errlog", dict(kmer_bias=5, fastq=True, emit_errlog=True)),
                 ("chimeric -k4 narrow", dict(kmer_bias=4, chimeric=True, min_len=2000, max_len=12000)),
                 ("narrow window (many passes)", dict(min_len=3000, max_len=9000, fastq=True))):
    p = E.make_params(seed=0xABCDEF01, first_read=0, n_reads=n, max_len=kw.pop("max_len", ref.max_chrom), meta=True, **kw)
    b = eng.generate(p)
    exp = O.generate_meta(mdl, ref, abun, infl if p.chimeric else None, p)
    try:
        compare(b, exp, p)
        assert np.array_equal(eng.species_bases(), exp["species_bases"])
        print("%-30s %6d reads  identical" % (name, n))
    except AssertionError as ex:
        bad += 1
        print("%-30s %6d reads  DIFFERS: %s" % (name, n, str(ex)[:200]))
sys.exit(1 if bad else 0)
