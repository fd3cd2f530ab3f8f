"""CPU-only: the host-compiled chain source (tests/chain_host.hip) against the oracle on the BENCH model (hg002-like, 10^6 training points), 1 500 pieces of
up to 60 000 bases per blob layout — every formulation that scripts/r05/chain_ab.sh times.  python scripts/chain_host_sweep.py"""
import sys, os, tempfile, time
sys.path.insert(0,'/root/repo')
from tests import test_chain_host as TC
from nanosim_amd import model as M, synth
L=TC._build('/tmp')
d=tempfile.mkdtemp(); prefix=os.path.join(d,'training')
synth.write_model(prefix, synth.SynthModelSpec(n_train=1_000_000, seed=20260926), write_pkl=False)
mdl=M.load_model(prefix)
t0=time.time()
for layout, variants in ((3,(32,31,30,33)),(1,(18,25,30,33)),(0,(0,2,11,17))):
    pk,n=TC.sweep(L, mdl, variants, 1500, 11+layout, (1,2,3,5,40,700,8000,20000,60000), layout=layout)
    print('layout',layout,'variants',variants,'events compared',n, 'whole',L.chost_whole(pk), round(time.time()-t0,1),'s'); sys.stdout.flush()
    L.chost_free(pk)
