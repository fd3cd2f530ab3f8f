"""An aligned worker call whose length window rejects about half of the reads per pass (-min_len 3000 -max_len 9000): per-call kernel times."""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as B  # noqa: E402
from nanosim_amd import engine, model, synth  # noqa: E402

tmp = tempfile.mkdtemp()
prefix = os.path.join(tmp, "hg002_like")
synth.write_model(prefix, synth.SynthModelSpec(n_train=1_000_000, seed=B.SEED), write_pkl=False)
mdl = model.load_model(prefix)
names, off, circ = B.reference_layout("ecoli")
ref = model.Reference(names, B.reference_bases("ecoli"), off, circ)
e = engine.Engine(0)
e.set_reference(ref)
e.load_model(mdl)
for i in range(5):
    b = e.generate(engine.make_params(seed=B.SEED, first_read=i * 1_000_000, n_reads=950_000, min_len=3000, max_len=9000))
    if i >= 2:
        print({k: round(v, 3) for k, v in b.kernel_ms().items() if v > 0.01}, "device_ms", round(float(b.info.ms_total), 3), "max attempts", int(b.reads()["attempts"].max()), flush=True)
e.close()
