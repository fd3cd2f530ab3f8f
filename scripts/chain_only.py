"""k_chain alone (no records): device time of the chain passes for the bench workload; used for A/B builds (NANOSIM_AMD_LIB)."""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nanosim_amd import engine as E, model, synth  # noqa: E402

SEED = 20260926


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    tmp = tempfile.mkdtemp(prefix="nschain_")
    prefix = os.path.join(tmp, "hg002_like")
    synth.write_model(prefix, synth.SynthModelSpec(n_train=1_000_000, seed=SEED), write_pkl=False)
    mdl = model.load_model(prefix)
    glen = synth.ECOLI_LEN
    seq = synth.synth_sequence(glen, SEED, n_frac=0.0005, iupac_frac=0.0002, lower_frac=0.02, hp_boost=0.005)
    ref = model.Reference(["ecoli-like"], seq, np.array([0, glen], dtype=np.uint64), np.array([1], dtype=np.uint8))
    e = E.Engine(0); e.set_reference(ref); e.load_model(mdl)
    ts = []
    for i in range(6):
        b = e.generate(E.make_params(seed=SEED, first_read=i * n, n_reads=n, max_len=ref.max_chrom, emit_records=False))
        ts.append(b.info.ms_kernel[1])
    print("k_chain ms", " ".join("%.3f" % t for t in ts))


main()
