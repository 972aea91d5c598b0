#!/bin/bash
# same-box A/B of library builds, eval renders only: tools/ab_eval.sh <lib.so|default> ...   (C2 eval, C5 full image)
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run this through gpurun}"
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for i in 1 2 3; do
for v in "$@"; do
  if [ "$v" = default ]; then unset CFNERF_LIB; else export CFNERF_LIB=$GRAFT_REPO_ROOT/$v; fi
  python bench.py --steps 40 --no-cpu-baseline --no-alt --mode eval 2>>gpurun_out/ab_bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v','C2 eval',round(d['value']),round(d['ms_per_step'],4), round(d['roofline']['launch_ms'],4))"
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt --config C5 2>>gpurun_out/ab_bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v','C5',round(d['value']),round(d['ms_per_step'],2))"
done
done
