"""The unaligned worker call of a bench step alone (50 000 reads on the ecoli-like reference, hg002_like model), K times: per-call kernel times."""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as B  # noqa: E402
from nanosim_amd import engine, model, synth  # noqa: E402

K = int(os.environ.get("K", "6"))
tmp = tempfile.mkdtemp()
prefix = os.path.join(tmp, "hg002_like")
synth.write_model(prefix, synth.SynthModelSpec(n_train=1_000_000, seed=B.SEED), write_pkl=False)
mdl = model.load_model(prefix)
names, off, circ = B.reference_layout("ecoli")
ref = model.Reference(names, B.reference_bases("ecoli"), off, circ)
e = engine.Engine(0)
if os.environ.get("BACKGROUND"):
    e.set_background(True)
e.set_reference(ref)
e.load_model(mdl)
for i in range(K + 2):
    b = e.generate(engine.make_params(seed=B.SEED, first_read=950_000 + i * 1_000_000, n_reads=50_000, kind=engine.NS_KIND_UNALIGNED, max_len=int(off[-1])))
    if i >= 2:
        print({k: round(v, 3) for k, v in b.kernel_ms().items()}, "seq_len max", int(b.reads()["seq_len"].max()), flush=True)
e.close()
