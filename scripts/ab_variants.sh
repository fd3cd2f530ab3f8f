#!/bin/bash
# A/B of prebuilt engine variants under nanosim_amd/_variants (profiling aid)
for f in nanosim_amd/_variants/*.so; do echo -n "$f "; NANOSIM_AMD_LIB=$PWD/$f python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), 'Mreads/s', {k: round(v,2) for k,v in d['kernel_ms'].items() if v>0.01})"; done
