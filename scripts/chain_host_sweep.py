"""CPU-only: the host-compiled chain source (tests/chain_host.hip) against the oracle on the BENCH model (hg002-like, 10^6 training points), 1 500 pieces of
up to 60 000 bases: chain_error_list on the LDS image, chain_error_list_g on the fp64 tables, chain_unaligned_error_list.  python scripts/chain_host_sweep.py"""
import sys, os, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_chain_host as TC
from nanosim_amd import model as M, synth
L = TC._build('/tmp')
d = tempfile.mkdtemp(); prefix = os.path.join(d, 'training')
synth.write_model(prefix, synth.SynthModelSpec(n_train=1_000_000, seed=20260926), write_pkl=False)
mdl = M.load_model(prefix)
t0 = time.time()
pk, n = TC.sweep(L, mdl, TC.ALL, 1500, 11, (1, 2, 3, 5, 40, 700, 8000, 20000, 60000))
print('variants', TC.ALL, 'events compared', n, 'whole', L.chost_whole(pk), 'LDS image', L.chost_lds_words(pk) * 8, 'bytes', round(time.time() - t0, 1), 's')
L.chost_free(pk)
