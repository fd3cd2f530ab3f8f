import os, sys, tempfile, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from nanosim_amd import engine as E, model as M, synth
d = tempfile.mkdtemp()
synth.write_model(d + "/training", synth.SynthModelSpec(n_train=1_000_000, seed=1), write_pkl=False)
mdl = M.load_model(d + "/training")
seq = synth.synth_sequence(synth.ECOLI_LEN, 1, n_frac=0.0005, iupac_frac=0.0002, lower_frac=0.02, hp_boost=0.005)
ref = M.Reference(["ecoli-like"], seq, np.array([0, len(seq)], dtype=np.uint64), np.array([1], dtype=np.uint8))
eng = E.Engine(0); eng.set_reference(ref); eng.load_model(mdl)
KIND = E.NS_KIND_UNALIGNED
def ok(first, n):
    try:
        eng.generate(E.make_params(seed=1, first_read=first, n_reads=n, max_len=ref.max_chrom, kind=KIND)); return True
    except E.EngineError as ex:
        return False
lo, n = 2850000, 150000
# find first failing 1M batch
for b in range(lo, lo + n, 1000000):
    m = min(1000000, lo + n - b)
    if not ok(b, m):
        lo, n = b, m
        break
else:
    print("no failure"); sys.exit()
while n > 1:
    h = n // 2
    if not ok(lo, h): n = h
    else: lo, n = lo + h, n - h
print("failing read", lo)
try:
    eng.generate(E.make_params(seed=1, first_read=lo, n_reads=1, max_len=ref.max_chrom, kind=KIND))
except E.EngineError as ex:
    print(ex)
b = eng.generate(E.make_params(seed=1, first_read=lo, n_reads=1, max_len=ref.max_chrom, emit_records=False, kind=E.NS_KIND_PERFECT))
print("perfect-mode length of the same read", b.reads()["seq_len"], b.pieces()["ref_len"])
