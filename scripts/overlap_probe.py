"""Does the chain of one batch overlap with the record stage of another?  Two engine contexts on one GPU, each generating half of the
aligned batch of a bench step from its own host thread, against one context generating the whole batch.
    python scripts/overlap_probe.py [reads] [stagger_ms]"""
import os
import sys
import tempfile
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nanosim_amd import engine as E, model, synth  # noqa: E402

SEED = 20260926


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 950000
    stagger = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
    tmp = tempfile.mkdtemp(prefix="nsprobe_")
    prefix = os.path.join(tmp, "hg002_like")
    synth.write_model(prefix, synth.SynthModelSpec(n_train=1_000_000, seed=SEED), write_pkl=False)
    mdl = model.load_model(prefix)
    glen = synth.ECOLI_LEN
    seq = synth.synth_sequence(glen, SEED, n_frac=0.0005, iupac_frac=0.0002, lower_frac=0.02, hp_boost=0.005)
    ref = model.Reference(["ecoli-like"], seq, np.array([0, glen], dtype=np.uint64), np.array([1], dtype=np.uint8))
    engs = [E.Engine(0) for _ in range(2)]
    for e in engs:
        e.set_reference(ref); e.load_model(mdl)

    def run(e, first, cnt, delay=0.0):
        if delay:
            time.sleep(delay)
        e.generate(E.make_params(seed=SEED, first_read=first, n_reads=cnt, max_len=glen))

    def step_one(i):
        run(engs[0], i * n, n)

    def step_two(i):
        h = n // 2
        t = threading.Thread(target=run, args=(engs[1], i * n + h, n - h, stagger * 1e-3))
        t.start(); run(engs[0], i * n, h); t.join()

    for name, fn in (("one context", step_one), ("two contexts, half each", step_two), ("one context", step_one), ("two contexts, half each", step_two)):
        for i in range(2):
            fn(i)
        t0 = time.perf_counter()
        for i in range(6):
            fn(2 + i)
        dt = (time.perf_counter() - t0) / 6
        print("%-26s %.2f ms per %d reads  (%.1f M reads/s)" % (name, dt * 1e3, n, n / dt / 1e6))


main()
