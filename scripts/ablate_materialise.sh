#!/bin/bash
# (build with -DNS_ABLATE: scripts/ab_build.sh ablate:"-DNS_ABLATE", NANOSIM_AMD_LIB=nanosim_amd/_variants/ablate.so — the bits are compiled out of the product build since round 6)
# k_materialise time per ablated phase (NS_DEBUG_SKIP bits: 1 the copy loop, 2 the letters, 8 head/tail, 16 the 16-byte stores,
# 64 no quality table look-ups, 128 no quality Philox, 256 no qualities).  Profiling aid only: results are wrong when a bit is set.
for d in ${@:-0 1 2 8 16}; do echo -n "skip=$d "; NS_DEBUG_SKIP=$d timeout 60 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['kernel_ms']['k_materialise'],2))"; done
