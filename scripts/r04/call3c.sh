#!/bin/bash
# round 4, GPU call 3c: bench lines of configs[4] (metagenome) and the CLI trace of a 3-step genome run (light on /dev/shm: 10^6 reads per step kept small)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04f; mkdir -p $O; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "errlog or equals_oracle" 2>&1 | tail -4 ) > $O/pytest_parity.log 2>&1; cat $O/pytest_parity.log
timeout 600 python bench.py --metagenome --no-e2e > $O/bench_zymo10_metagenome.json 2> $O/bench_zymo10.err; cut -c1-300 $O/bench_zymo10_metagenome.json; tail -2 $O/bench_zymo10.err
( NS_CLI_TRACE=1 timeout 300 python scripts/bench_cli.py -n 2000000 ) > $O/bench_cli_default.log 2>&1; tail -8 $O/bench_cli_default.log
