set -uo pipefail
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r06_soak2.txt"; cd "$R"; : > "$O"
run() { echo "## $1" >> "$O"; shift; "$@" 2>&1 | tail -3 >> "$O"; }
run fuzz_300_420 env CFNERF_FUZZ_SEEDS=300-420 python -m pytest tests/test_hip_train.py -q -m gpu -k random_conf
run fuzz_1040_1100 env CFNERF_FUZZ_SEEDS=1040-1100 python -m pytest tests/test_hip_train.py -q -m gpu -k random_conf
run fuzz_2040_2100 env CFNERF_FUZZ_SEEDS=2040-2100 python -m pytest tests/test_hip_train.py -q -m gpu -k random_conf
run fuzz_3016_3030 env CFNERF_FUZZ_SEEDS=3016-3030 python -m pytest tests/test_hip_train.py -q -m gpu -k random_conf
run fuzz_b16_130_170 env CFNERF_FUZZ_SEEDS=130-170 CFNERF_FUZZ_PREC=bf16x3 python -m pytest tests/test_hip_train.py -q -m gpu -k random_conf
run seam_200_400 env CFNERF_FUZZ_SEEDS=200-400 python -m pytest tests/test_hip_unfused_seam.py -q -m gpu -k random_shapes
cat "$O"
