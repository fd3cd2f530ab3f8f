#!/bin/bash
# round 4, GPU call 2b: the transcriptome block walk on the GPU
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04c; mkdir -p $O; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 400 python -m pytest tests/test_gpu_transcriptome.py -m gpu -x -q 2>&1 | tail -25 ) > $O/pytest_trx.log 2>&1; cat $O/pytest_trx.log
( timeout 400 python -m pytest tests/test_gpu_cli.py -m gpu -x -q -k "transcriptome or two_context" 2>&1 | tail -25 ) > $O/pytest_cli_trx.log 2>&1; cat $O/pytest_cli_trx.log
( timeout 400 python scripts/parity_trx_big.py 20000 2>&1 | tail -8 ) > $O/parity_trx_big.log 2>&1; cat $O/parity_trx_big.log
( timeout 200 python scripts/bench_transcriptome.py; timeout 200 python scripts/bench_transcriptome.py --model-ir ) > $O/bench_transcriptome.log 2>&1; cat $O/bench_transcriptome.log
