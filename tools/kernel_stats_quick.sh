#!/bin/bash
# rocprofv3 --kernel-trace --stats of the default bench (C2 train, fp32) and the per-kernel averages, for a quick look between
# builds.  Run on the GPU box through gpurun:  bash tools/kernel_stats_quick.sh
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run this through gpurun (GRAFT_REPO_ROOT is the root of the repo copy on the GPU box)}"
R="$GRAFT_REPO_ROOT"
O="$R/gpurun_out"
[ -f "$R/bench.py" ] || { echo "no bench.py under $R" >&2; exit 1; }
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/q_train" -o q -- python3 "$R/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-alt > "$O/q_train.log" 2>&1
python3 - "$O/q_train/q_kernel_stats.csv" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print(f'{r["Name"][:50]:50s} calls={r["Calls"]:>5s} avg_us={float(r["AverageNs"]) / 1e3:9.1f}')
PY
