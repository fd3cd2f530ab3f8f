cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_metagenome.py tests/test_gpu_transcriptome.py -x -q -m gpu 2>&1 | tail -4
python bench.py --no-cpu-baseline --serial 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), round(d['ms_per_step'],2), d['aligned_batch']['kernel_ms'], d['unaligned_batch']['kernel_ms'])"
