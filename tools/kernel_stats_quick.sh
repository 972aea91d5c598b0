cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/q_train -o q -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-alt > $R/gpurun_out/q_train.log 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/q_train/q_kernel_stats.csv")))
for r in rows[:14]:
    print(f'{r["Name"][:50]:50s} calls={r["Calls"]:>5s} avg_us={float(r["AverageNs"])/1e3:9.1f}')
PY
