"""The record kernel of a process runs in one of two states (~4.7 or ~5.0 ms per 950 000 reads): one process per line — its record kernel's
own event pair over 6 steps (aligned only) and the device addresses of the buffers it reads and writes."""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import bench as B  # noqa: E402
from nanosim_amd import engine, model, synth  # noqa: E402

tmp = os.environ.get("PROBE_TMP") or tempfile.mkdtemp()
prefix = os.path.join(tmp, "hg002_like")
if not os.path.exists(prefix + "_kde.npz"):
    synth.write_model(prefix, synth.SynthModelSpec(n_train=1_000_000, seed=B.SEED), write_pkl=False)
mdl = model.load_model(prefix)
names, off, circ = B.reference_layout("ecoli")
ref = model.Reference(names, B.reference_bases("ecoli"), off, circ)
e = engine.Engine(0)
e.set_reference(ref)
e.load_model(mdl)
ms = []
for i in range(8):
    b = e.generate(engine.make_params(seed=B.SEED, first_read=i * 1_000_000, n_reads=950_000, max_len=int(off[-1])))
    if i >= 2:
        ms.append(float(b.info.ms_kernel[6]))
ptr = {nm: int(e.L.ns_device_ptr(e.ctx, k) or 0) for nm, k in (("records", 0), ("reads", 1), ("pieces", 2), ("events", 3))}
print("record kernel %.3f ms (min %.3f max %.3f)" % (np.mean(ms), min(ms), max(ms)), " ".join("%s=%#x" % kv for kv in ptr.items()), flush=True)
e.close()
