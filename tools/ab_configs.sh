#!/bin/bash
# same-box A/B of library builds over several configs:  tools/ab_configs.sh <lib.so|default> ...   (AB_CONFIGS="C2 C4 W512")
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run this through gpurun}"
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for i in 1 2; do
for v in "$@"; do
  if [ "$v" = default ]; then unset CFNERF_LIB; else export CFNERF_LIB=$GRAFT_REPO_ROOT/$v; fi
  for c in ${AB_CONFIGS:-C2 C4 W512}; do
    python "$GRAFT_REPO_ROOT/bench.py" --config $c --steps 30 --no-cpu-baseline --no-alt 2>>gpurun_out/ab_bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v','$c',round(d['value']),round(d['ms_per_step'],3),{k:round(x,3) for k,x in d['kernel_ms'].items()})"
  done
done
done
