cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -5
bash scripts/ab_run.sh --fastq --kmer-bias 5
