#!/bin/bash
# the bench lines of one round, one file per config (run on the GPU box through gpurun):  tools/bench_round.sh r05
set -uo pipefail
: "${GRAFT_REPO_ROOT:?run this through gpurun}"
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"; T="${1:-r05}"; mkdir -p "$O"
cd "$R"
python bench.py > "$O/bench_${T}_default.json" 2> "$O/bench_${T}_default.err"
B="--no-cpu-baseline --no-alt"
python bench.py $B --steps 50 > "$O/bench_${T}_c2.json" 2>> "$O/bench_${T}.err"
python bench.py $B --steps 50 --mode eval > "$O/bench_${T}_eval.json" 2>> "$O/bench_${T}.err"
python bench.py $B --steps 20 --config C3 > "$O/bench_${T}_c3.json" 2>> "$O/bench_${T}.err"
python bench.py $B --steps 50 --config C4 > "$O/bench_${T}_c4.json" 2>> "$O/bench_${T}.err"
python bench.py $B --steps 30 --config W512 > "$O/bench_${T}_w512.json" 2>> "$O/bench_${T}.err"
python bench.py $B --steps 50 --config C1 > "$O/bench_${T}_c1.json" 2>> "$O/bench_${T}.err"
python bench.py $B --steps 3 --warmup 1 --config C5 > "$O/bench_${T}_c5.json" 2>> "$O/bench_${T}.err"
python bench.py $B --steps 30 --config K64 > "$O/bench_${T}_k64.json" 2>> "$O/bench_${T}.err"
python bench.py $B --steps 10 --warmup 3 --config N8192 > "$O/bench_${T}_n8192.json" 2>> "$O/bench_${T}.err"
# the N > 1 line with everything in it (psnr + vs_single_process, comm, rank_skew, cpu_baseline): two ranks on ONE GPU over gloo, a code-path run
CFNERF_BENCH_SAME_GPU=1 python bench.py --gpus 2 --steps 20 --psnr-steps 500 > "$O/bench_${T}_2ranks_same_gpu.json" 2>> "$O/bench_${T}.err"
for f in "$O"/bench_${T}_*.json; do python - "$f" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split("/")[-1], round(d["value"]), round(d["ms_per_step"], 3), "frac", round(d["roofline"]["frac"], 3), d.get("kernel_ms"), "step_frac", d.get("step_frac_of_peak"))
PY
done
