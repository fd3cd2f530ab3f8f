"""Unaligned reads (simulation_unaligned, S:1482-1549) of the bench workload: device time per batch."""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nanosim_amd import engine as E, model, synth  # noqa: E402

SEED = 20260926


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
    fastq = len(sys.argv) > 2 and sys.argv[2] == "fastq"
    tmp = tempfile.mkdtemp(prefix="nsun_")
    prefix = os.path.join(tmp, "hg002_like")
    synth.write_model(prefix, synth.SynthModelSpec(n_train=1_000_000, seed=SEED), write_pkl=False)
    mdl = model.load_model(prefix, fastq=fastq)
    glen = synth.ECOLI_LEN
    seq = synth.synth_sequence(glen, SEED, n_frac=0.0005, iupac_frac=0.0002, lower_frac=0.02, hp_boost=0.005)
    ref = model.Reference(["ecoli-like"], seq, np.array([0, glen], dtype=np.uint64), np.array([1], dtype=np.uint8))
    e = E.Engine(0); e.set_reference(ref); e.load_model(mdl)
    for i in range(4):
        b = e.generate(E.make_params(seed=SEED, first_read=i * n, n_reads=n, kind=E.NS_KIND_UNALIGNED, max_len=ref.max_chrom, fastq=fastq))
        print("reads", n, "mean len %.0f" % (int(b.info.total_bases) / n), "events/read %.0f" % (int(b.info.events_used) / n),
              "ms total %.3f" % b.info.ms_total, " ".join("%s=%.3f" % (k, v) for k, v in zip(E.KERNEL_NAMES, b.info.ms_kernel)))


main()
