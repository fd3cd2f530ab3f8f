"""Probe: is there time to win by running the chain of one half-batch under the record kernel of the other?
Two independent engine contexts on one GPU, a Python thread each (the C call releases the GIL), the second one started `lag` ms late.
  one     : one context, K calls of 950 000 aligned reads
  halves  : two contexts, K calls of 475 000 each, side by side at several lags        -> ms per 950 000
  steps   : two contexts, K calls of 950 000 each, side by side at several lags        -> ms per 950 000
"""
import os
import sys
import tempfile
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as B  # noqa: E402
from nanosim_amd import engine, model, synth  # noqa: E402

K = int(os.environ.get("K", "8"))
N = 950_000


def main():
    tmp = tempfile.mkdtemp()
    prefix = os.path.join(tmp, "hg002_like")
    synth.write_model(prefix, synth.SynthModelSpec(n_train=1_000_000, seed=B.SEED), write_pkl=False)
    mdl = model.load_model(prefix)
    names, off, circ = B.reference_layout("ecoli")
    ref = model.Reference(names, B.reference_bases("ecoli"), off, circ)
    engs = [engine.Engine(0), engine.Engine(0)]
    for e in engs:
        e.set_reference(ref)
        e.load_model(mdl)
    max_len = int(off[-1])

    def run(e, n, base, k, lag, bar):
        bar.wait()
        if lag:
            time.sleep(lag * 1e-3)
        for i in range(k):
            e.generate(engine.make_params(seed=B.SEED, first_read=base + i * N, n_reads=n, max_len=max_len))

    def timed(specs, k):
        bar = threading.Barrier(len(specs) + 1)
        th = [threading.Thread(target=run, args=(e, n, base, k, lag, bar)) for e, n, base, lag in specs]
        for t in th:
            t.start()
        bar.wait()
        t0 = time.perf_counter()
        for t in th:
            t.join()
        return (time.perf_counter() - t0) * 1e3

    timed([(engs[0], N, 0, 0)], 2); timed([(engs[1], N, 0, 0)], 2)
    for rep in range(2):
        print("one      %.2f ms per 950 000" % (timed([(engs[0], N, 0, 0)], K) / K), flush=True)
        for lag in (0, 1.5, 2.5, 3.5):
            t = timed([(engs[0], N // 2, 0, 0), (engs[1], N // 2, N // 2, lag)], K)
            print("halves   lag %.1f  %.2f ms per 950 000" % (lag, t / K), flush=True)
        for lag in (0, 2.5, 4.0, 5.5):
            t = timed([(engs[0], N, 0, 0), (engs[1], N, 1 << 30, lag)], K)
            print("steps    lag %.1f  %.2f ms per 950 000" % (lag, t / (2 * K)), flush=True)
    for e in engs:
        e.close()


if __name__ == "__main__":
    main()
