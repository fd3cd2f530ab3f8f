#!/bin/bash
# round 4, first GPU call: the CLI's two-context schedule + bench.py's new entry points, the I/O probe of this box
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04a; mkdir -p $O; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 900 python -m pytest tests/test_gpu_cli.py tests/test_gpu_sinks.py -m gpu -x -q 2>&1 | tail -15 ) > $O/pytest_cli.log 2>&1
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
NS_BENCH_DEVICE=0 timeout 300 python bench.py --gpus 2 --dist-backend gloo --steps 3 --warmup 1 > $O/bench_2ranks_one_gpu_gloo.log 2>&1
./scripts/microbench/io_probe sys > $O/io_probe.log 2>&1
timeout 300 ./scripts/microbench/io_probe /dev/shm 4 >> $O/io_probe.log 2>&1
mkdir -p /root/ns_io_probe && timeout 200 ./scripts/microbench/io_probe /root/ns_io_probe 2 >> $O/io_probe.log 2>&1
( NS_CLI_TRACE=1 timeout 300 python scripts/bench_cli.py -n 3000000 ) > $O/bench_cli_default.log 2>&1
( NS_KEEP_SUBFILES=1 timeout 300 python scripts/bench_cli.py -n 3000000 -t 16 ) > $O/bench_cli_t16_keep.log 2>&1
tail -3 $O/pytest_cli.log; cut -c1-300 $O/bench_default.json; tail -2 $O/bench_cli_default.log $O/bench_cli_t16_keep.log
