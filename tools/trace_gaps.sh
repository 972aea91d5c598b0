#!/bin/bash
# kernel trace of a few train steps and the idle gaps between consecutive kernels of a step (run through gpurun)
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run this through gpurun}"
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rm -rf "$O/tg"
rocprofv3 --kernel-trace --output-format csv -d "$O/tg" -o tg -- python3 "$R/bench.py" --steps 6 --warmup 3 --no-cpu-baseline --no-alt ${TRACE_ARGS:-} > "$O/tg.log" 2>&1
python3 - "$O/tg" <<'PY'
import csv, glob, sys, os
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# last full step: from the last-but-one fused_fwd train launch to the next one
idx = [i for i, r in enumerate(rows) if "fused_fwd_kernel" in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = None
tot_busy = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("cfnerf::", "")[:60]
    print(f"{(s - t0) / 1e3:9.1f} us  +gap {gap:6.1f}  dur {(e - s) / 1e3:8.1f}  {name}")
    prev_end = max(prev_end or 0, e); tot_busy += e - s
print("step span", (int(rows[b]["Start_Timestamp"]) - t0) / 1e3, "us; sum of kernel durations", tot_busy / 1e3)
PY
