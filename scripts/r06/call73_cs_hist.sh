#!/bin/bash
# round 6, GPU call 73: k_cs_hist with an 8-byte register window per thread and without the second walk of every string (build_ab/libns_prev.so: the library before)
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06cs; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 900 python -m pytest tests/test_gpu_zz_characterize.py -m gpu -x -q 2>&1 | tail -3 ) | tee $O/pytest.log
for i in 1 2; do
  echo "== before" | tee -a $O/ab.log; NANOSIM_AMD_LIB=$PWD/build_ab/libns_prev.so timeout 600 python scripts/bench_characterize.py 2>&1 | tail -1 | tee -a $O/ab.log
  echo "== now" | tee -a $O/ab.log; timeout 600 python scripts/bench_characterize.py 2>&1 | tail -1 | tee -a $O/ab.log
done
