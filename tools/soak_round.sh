#!/bin/bash
# soak of a round's library (run through gpurun):  tools/soak_round.sh <tag>  ->  gpurun_out/<tag>_soak.txt
#   fuzz draws of tests/test_hip_train.py (fp32: several seed families; bf16x3), random shapes of the unfused seam, bit-reproducibility, leak check
set -uo pipefail
: "${GRAFT_REPO_ROOT:?run this through gpurun}"
R="$GRAFT_REPO_ROOT"; T="${1:-r06}"; O="$R/gpurun_out/${T}_soak.txt"; cd "$R"; mkdir -p gpurun_out; : > "$O"
run() { echo "## $1" >> "$O"; shift; "$@" 2>&1 | tail -4 >> "$O"; }
run repro_soak python tests/tools/repro_soak.py
run leak python tests/tools/leak_check.py
run fuzz_100_220 env CFNERF_FUZZ_SEEDS=100-220 python -m pytest tests/test_hip_train.py -q -m gpu -k random_conf
run fuzz_1000_1040 env CFNERF_FUZZ_SEEDS=1000-1040 python -m pytest tests/test_hip_train.py -q -m gpu -k random_conf
run fuzz_2000_2040 env CFNERF_FUZZ_SEEDS=2000-2040 python -m pytest tests/test_hip_train.py -q -m gpu -k random_conf
run fuzz_3000_3016 env CFNERF_FUZZ_SEEDS=3000-3016 python -m pytest tests/test_hip_train.py -q -m gpu -k random_conf
run fuzz_5000_5031 env CFNERF_FUZZ_SEEDS=5000-5031 python -m pytest tests/test_hip_train.py -q -m gpu -k random_conf
run fuzz_b16_100_130 env CFNERF_FUZZ_SEEDS=100-130 CFNERF_FUZZ_PREC=bf16x3 python -m pytest tests/test_hip_train.py -q -m gpu -k random_conf
run seam_100_160 env CFNERF_FUZZ_SEEDS=100-160 python -m pytest tests/test_hip_unfused_seam.py -q -m gpu -k random_shapes
cat "$O"
