#!/bin/bash
# round 5, GPU call 6: k_errlog<BUF> (block buffer by the batch's row length, lane groups, prefetched event) — parity, then A/B of the
# register bound (six wavefronts per SIMD with 20 bytes of scratch against five without)
cd "$(dirname "$0")/../.."
O=gpurun_out/r05g; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_metagenome.py tests/test_gpu_transcriptome.py -m gpu -x -q 2>&1 | tail -5 ) | tee $O/pytest_parity.log
for v in w6 w1 w6 w1; do echo -n "$v "; NANOSIM_AMD_LIB=$PWD/nanosim_amd/_variants/$v.so timeout 100 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --aligned-only --errlog 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:round(v,3) for k,v in d['kernel_ms'].items() if v>0.01})"; done | tee $O/ab_errlog_minw.log
echo -n "w6 large buffer "; NS_ERRLOG_BUF_LARGE=1 NANOSIM_AMD_LIB=$PWD/nanosim_amd/_variants/w6.so timeout 100 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --aligned-only --errlog 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:round(v,3) for k,v in d['kernel_ms'].items() if v>0.01})" | tee -a $O/ab_errlog_minw.log
echo -n "w6 metagenome "; NANOSIM_AMD_LIB=$PWD/nanosim_amd/_variants/w6.so timeout 100 python bench.py --metagenome --steps 3 --warmup 2 --no-cpu-baseline --aligned-only --errlog 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:round(v,3) for k,v in d['kernel_ms'].items() if v>0.01})" | tee -a $O/ab_errlog_minw.log
