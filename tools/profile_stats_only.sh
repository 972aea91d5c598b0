#!/bin/bash
# only the kernel-trace + stats passes of tools/profile_round.sh (no counter passes):  tools/profile_stats_only.sh
set -uo pipefail
: "${GRAFT_REPO_ROOT:?run this through gpurun}"
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-alt"
stats() { local n="$1"; shift; rm -rf "$O/i_$n"; rocprofv3 --kernel-trace --stats --output-format csv -d "$O/i_$n" -o "$n" -- python3 "$R/bench.py" --steps 50 --warmup 3 $B "$@" > "$O/i_$n.log" 2>&1; }
stats train
stats eval --mode eval
stats train_b16 --precision bf16x3
stats w512 --config W512
stats c4 --config C4
stats c3 --config C3
stats k64 --config K64
find "$O"/i_* -name "*kernel_trace.csv" -size +1M -delete
for n in train eval w512 k64; do head -3 "$O/i_$n/${n}_kernel_stats.csv" | cut -c1-160; done
