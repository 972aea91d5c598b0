#!/bin/bash
# per-job cost of the small weight-gradient launch (round 5): the experiment build cf-nerf_amd/libvar_dsonly.so
# (profiles/patches/r05_dw_small_only_job.patch applied, tools/build_variant.sh dsonly) keeps the small tiles of ONE job when
# CFN_DS_ONLY=<add_job index> is set; rocprofv3 --kernel-trace --stats gives dw_small_kernel's own duration for each.
#   gpurun -- bash tools/dw_small_jobs.sh [bench args, default: --config C2]
set -uo pipefail
: "${GRAFT_REPO_ROOT:?run this through gpurun}"
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r5/dsjobs"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
export CFNERF_LIB="$R/cf-nerf_amd/libvar_dsonly.so"
ARGS="${*:---config C2}"
for job in all 0 5 9 12 13 14 15; do
  if [ "$job" = all ]; then unset CFN_DS_ONLY; else export CFN_DS_ONLY=$job; fi
  rm -rf "$O/j_$job"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/j_$job" -o t -- python3 "$R/bench.py" --no-cpu-baseline --no-alt --steps 20 --warmup 3 $ARGS > "$O/j_$job.log" 2>&1
  f=$(find "$O/j_$job" -name "t_kernel_stats.csv" | head -1)
  echo "job $job: $(grep -m1 'CFN_DS_ONLY' "$O/j_$job.log" | cut -c1-200)"
  [ -n "$f" ] && grep -E "dw_small|dw_big|reduce_weights" "$f" | awk -F, '{printf "   %-40s calls %s avg_ns %s\n", substr($1,1,40), $2, $4}'
done
