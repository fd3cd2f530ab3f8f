"""Device time per batch of the worker kinds the headline bench does not cover, on the bench workload (E. coli-like reference,
hg002-like model): unaligned reads (simulation_unaligned, S:1482-1549), chimeric reads (S:1276-1299, 1406-1419), --perfect.
    python scripts/bench_kinds.py unaligned 50000 | chimeric 1000000 | perfect 1000000   [fastq]"""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nanosim_amd import engine as E, model, synth  # noqa: E402

SEED = 20260926


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "unaligned"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
    fastq = len(sys.argv) > 3 and sys.argv[3] == "fastq"
    max_len = int(sys.argv[4]) if len(sys.argv) > 4 else None          # optional -max (the longest read of a batch sets the tail of the wave-per-read kernels)
    tmp = tempfile.mkdtemp(prefix="nskind_")
    prefix = os.path.join(tmp, "hg002_like")
    synth.write_model(prefix, synth.SynthModelSpec(n_train=1_000_000, seed=SEED), write_pkl=False)
    mdl = model.load_model(prefix, fastq=fastq, chimeric=what == "chimeric", perfect=what == "perfect")
    glen = synth.ECOLI_LEN
    seq = synth.synth_sequence(glen, SEED, n_frac=0.0005, iupac_frac=0.0002, lower_frac=0.02, hp_boost=0.005)
    ref = model.Reference(["ecoli-like"], seq, np.array([0, glen], dtype=np.uint64), np.array([1], dtype=np.uint8))
    e = E.Engine(0); e.set_reference(ref); e.load_model(mdl)
    kind = {"unaligned": E.NS_KIND_UNALIGNED, "perfect": E.NS_KIND_PERFECT}.get(what, E.NS_KIND_ALIGNED)
    for i in range(4):
        b = e.generate(E.make_params(seed=SEED, first_read=i * n, n_reads=n, kind=kind, chimeric=what == "chimeric", max_len=max_len or ref.max_chrom, fastq=fastq))
        print(what, "reads", n, "mean len %.0f" % (int(b.info.total_bases) / n), "events/read %.0f" % (int(b.info.events_used) / n),
              "pieces/read %.2f" % (int(b.info.n_pieces) / n), "ms total %.3f" % b.info.ms_total,
              " ".join("%s=%.3f" % (k, v) for k, v in zip(E.KERNEL_NAMES, b.info.ms_kernel)))


main()
