#!/bin/bash
# generic one-off PMC probe of a bench workload (through gpurun):  PB_ARGS="--mode train" PB_KERNEL=fused_fwd bash tools/pmc_probe.sh "CTR_A CTR_B" "CTR_C ..."
# one rocprofv3 pass per quoted counter group (counters only with --kernel-trace), mean per dispatch of the kernels matching PB_KERNEL
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run this through gpurun}"
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
set +e
rm -rf "$O"/q_*
i=0
for c in "$@"; do
  # (bounded: a counter group the hardware cannot collect in one pass makes rocprofv3 abort and then sit in its finaliser - that cost
  #  a 25-minute gpurun call in round 4; FETCH_SIZE / WRITE_SIZE go in passes of their own, like tools/profile_round.sh does)
  timeout 420 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$O/q_$i" -o pmc -- python3 "$R/bench.py" --no-cpu-baseline --no-alt --steps 3 --warmup 1 ${PB_ARGS:-} > "$O/q_$i.log" 2>&1
  i=$((i+1))
done
python3 - "$O" "${PB_KERNEL:-cfnerf}" <<'PY'
import csv, glob, os, sys
O, key = sys.argv[1], sys.argv[2]
for d in sorted(glob.glob(os.path.join(O, "q_*"))):
    if not os.path.isdir(d): continue
    f = glob.glob(os.path.join(d, "**", "pmc_counter_collection.csv"), recursive=True)
    if not f: print(d, "no csv"); continue
    acc = {}
    for r in csv.DictReader(open(f[0])):
        if key not in r["Kernel_Name"]: continue
        k = r["Kernel_Name"].split("(")[0][-48:]
        acc.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(os.path.basename(d), k, {c: round(sum(x) / len(x)) for c, x in v.items()})
PY
