#!/bin/bash
# same-box A/B of library builds: tools/ab_bench.sh <lib.so|default> ...   (eval + train, fp32)
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run this through gpurun}"
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for i in 1 2; do
for v in "$@"; do
  if [ "$v" = default ]; then unset CFNERF_LIB; else export CFNERF_LIB=$GRAFT_REPO_ROOT/$v; fi
  for m in ${AB_MODES:-fp32}; do
    pm=""; [ $m = bf16x3 ] && pm="--precision bf16x3"
    python "$GRAFT_REPO_ROOT/bench.py" --steps 30 --no-cpu-baseline --no-alt $pm 2>>gpurun_out/ab_bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v','train','$m',round(d['value']),round(d['ms_per_step'],3),d['kernel_ms'])"
    python "$GRAFT_REPO_ROOT/bench.py" --steps 30 --no-cpu-baseline --no-alt --mode eval $pm 2>>gpurun_out/ab_bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v','eval','$m',round(d['value']),round(d['ms_per_step'],3))"
  done
done
done
