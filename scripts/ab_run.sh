#!/bin/bash
# A/B of the prebuilt variants under nanosim_amd/_variants (scripts/ab_build.sh): scripts/ab_run.sh [bench.py arguments]; NS_DEBUG_SKIP list in $SKIPS;
# VARIANTS="a b" limits the run to those names
cd $GRAFT_REPO_ROOT
for f in nanosim_amd/_variants/*.so; do
  name=$(basename $f .so)
  if [ -n "$VARIANTS" ]; then case " $VARIANTS " in *" $name "*) ;; *) continue;; esac; fi
  for d in ${SKIPS:-0}; do echo -n "$name skip=$d "; NS_DEBUG_SKIP=$d NANOSIM_AMD_LIB=$PWD/$f timeout 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-genome-run "$@" 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), 'Mreads/s', {k: round(v,2) for k,v in d['kernel_ms'].items() if v>0.01})"; done; done
