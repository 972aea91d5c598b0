#!/bin/bash
# build a VARIANT of the library for same-box A/B runs (tools/ab_kernels.sh, CFNERF_LIB=...) from the CURRENT tree - e.g. with an
# experiment applied as a patch, or extra compiler flags (the concluded CFN_* source switches of rounds 2-4 are gone: profiles/EXPERIMENTS.md):
#   tools/build_variant.sh <name> [compiler flags ...]   ->  cf-nerf_amd/libvar_<name>.so   (git-ignored, travels with gpurun)
set -euo pipefail
R="$(cd "$(dirname "$0")/.." && pwd)"
name="$1"; shift
O="$R/cf-nerf_amd/build/var_$name"; mkdir -p "$O"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Wno-unused-result -Wno-unused-value"
for f in cfnerf_fwd cfnerf_bwd cfnerf_abi; do
  hipcc $FLAGS "$@" -c "$R/cf-nerf_amd/csrc/$f.hip" -o "$O/$f.o" &
done
hipcc $FLAGS -fno-slp-vectorize "$@" -c "$R/cf-nerf_amd/csrc/cfnerf_tail.hip" -o "$O/cfnerf_tail.o" &      # (per-file flag: see cf-nerf_amd/build.py)
wait
hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script="$R/cf-nerf_amd/csrc/cfnerf_exports.map" -o "$R/cf-nerf_amd/libvar_$name.so" "$O"/cfnerf_fwd.o "$O"/cfnerf_bwd.o "$O"/cfnerf_tail.o "$O"/cfnerf_abi.o
echo "$R/cf-nerf_amd/libvar_$name.so"
