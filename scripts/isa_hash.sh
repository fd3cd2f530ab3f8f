#!/bin/bash
# md5 of the device ISA of the engine (hipcc -S, device pass only, without the per-compilation unit id): two builds with the same hash run
# the same kernels.  scripts/isa_hash.sh [-D flags...]   Round 4 used it to show that the refactors made after the GPU budget was closed
# (ns_pack.h, NS_DEV, the flagged formulations of that round at 0) left the GPU-verified kernels byte-identical: profiles/r04/README.md.
cd "$(dirname "$0")/.."
out=$(mktemp /tmp/ns_isa_XXXXXX.s)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --cuda-device-only "$@" -S -o $out nanosim_amd/csrc/nanosim_amd.hip 2>/dev/null || exit 1
grep -v "__hip_cuid_" $out | md5sum | cut -d' ' -f1
rm -f $out
