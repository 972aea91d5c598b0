#!/bin/bash
# alternating same-box A/B (three rounds) with per-stage medians over 200 steps:  tools/ab_kernels.sh <lib.so|default> ...   (AB_ARGS="--config C2")
set -uo pipefail
: "${GRAFT_REPO_ROOT:?run this through gpurun}"
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for i in 1 2 3; do
for v in "$@"; do
  if [ "$v" = default ]; then unset CFNERF_LIB; else export CFNERF_LIB=$GRAFT_REPO_ROOT/$v; fi
  python tools/ab_kernels.py ${AB_ARGS:-} 2>>gpurun_out/ab_kernels.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['lib'], d['config'], d['mode'], 'step', d['step_ms']['median'], ' '.join(f\"{k} {v['median']}/{v['p10']}\" for k, v in d.items() if isinstance(v, dict) and k != 'step_ms'))"
done
done
