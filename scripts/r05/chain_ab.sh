#!/bin/bash
# round 5, first GPU call: the formulations of the event chain prepared in round 4 (ns_chain.h, NS_CHAIN_VAR) against the product build.
#   here (no GPU):  scripts/r05/chain_ab.sh build      -> nanosim_amd/_variants/*.so (they travel to the GPU box with the snapshot)
#   on the GPU box: scripts/r05/chain_ab.sh            -> per variant: the bit-exact parity tests, then 3 bench steps with the kernel times
# Every variant is held against the oracle on the CPU by tests/test_chain_host.py (same events, bit for bit); what the GPU adds is the
# timing and the parity of the whole batch through k_chain.
#   var1  one ev_push32 site            var2  run_length_w            var4  column looked up before the event store
#   var8  run-length record in LDS instead of three kernel-argument loads per event (blob layout 1)
#   v2    chain_error_list_v2: one-word ECDF segments, column word, reads issued round by round in one basic block (blob layout 3)
#   bitop3   the three-way XORs of a Philox round as one v_bitop3_b32 each (ns_rng.h): -20 vector instructions per evaluation, every kernel
#   rekey    the Philox key made opaque per evaluation (ns_rng.h): the twenty round-key words are recomputed by the scalar unit instead of
#            living in SGPRs across the kernels' loops — statically -109 SGPR-spill accesses in the record kernel, -106 in k_chain (whose
#            VGPR count falls from 104 to 97; with v2 + bitop3 to 94 = five wavefronts per SIMD without forcing them)
#   errlog3  k_errlog with the packed row writer (ns_errlog.h: errlog_tail_v3; tests/test_errlog_host.py); all = v2 + errlog3 + bitop3 + rekey
#   v2m5  v2 compiled for five wavefronts per SIMD (96 VGPRs, 20 bytes of spills outside the loop): the bench model's LDS image is 24.1 KB
#         in layout 3 (29.1 KB in layout 0), + 8 KB of event staging = five workgroups of 256 threads per CU instead of four
cd "$(dirname "$0")/../.."
if [ "$1" = build ]; then
  exec scripts/ab_build.sh base:"" var1:"-DNS_CHAIN_VAR=1" var2:"-DNS_CHAIN_VAR=2" var4:"-DNS_CHAIN_VAR=4" var8:"-DNS_CHAIN_VAR=8" var9:"-DNS_CHAIN_VAR=9" \
       var11:"-DNS_CHAIN_VAR=11" v2:"-DNS_CHAIN_VAR=40" v2m5:"-DNS_CHAIN_VAR=40 -DNS_CHAIN_MINW=5" \
       errlog3:"-DNS_ERRLOG_V3" bitop3:"-DNS_PHILOX_BITOP3" rekey:"-DNS_PHILOX_REKEY" \
       all:"-DNS_CHAIN_VAR=40 -DNS_ERRLOG_V3 -DNS_PHILOX_BITOP3 -DNS_PHILOX_REKEY"
fi
O=gpurun_out/r05a; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for f in nanosim_amd/_variants/*.so; do
  name=$(basename $f .so)
  echo "== $name parity"; ( NANOSIM_AMD_LIB=$PWD/$f timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 ) | tee $O/parity_$name.log
done
scripts/ab_run.sh 2>&1 | tee $O/ab_chain.log
for name in base errlog3 all; do            # the error profile switched on: what k_errlog costs (bench.py's errlog_on object)
  echo -n "$name errlog_on "; NANOSIM_AMD_LIB=$PWD/nanosim_amd/_variants/$name.so timeout 300 python bench.py --no-cpu-baseline --no-e2e --no-configs2 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); e=d['errlog_on']; print(round(e['value']/1e6,1), 'M reads/s', e['ms_per_step'], 'ms/step, k_errlog', e['k_errlog_ms'], 'ms')"
done 2>&1 | tee $O/ab_errlog.log
( timeout 300 python -m pytest tests/test_gpu_zz_characterize.py -m gpu -q 2>&1 | tail -5 ) | tee $O/pytest_characterize.log
