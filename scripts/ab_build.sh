#!/bin/bash
# builds engine variants for an A/B on the GPU box: scripts/ab_build.sh name1:"-DX=1" name2:"-DY=2 -DZ" ...  -> nanosim_amd/_variants/<name>.so
cd "$(dirname "$0")/.."
mkdir -p nanosim_amd/_variants; rm -f nanosim_amd/_variants/*.so
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared $flags -o nanosim_amd/_variants/$name.so nanosim_amd/csrc/nanosim_amd.hip 2>&1 | grep -E "error" ) &
done
wait; ls -la nanosim_amd/_variants/
