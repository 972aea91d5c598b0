#!/bin/bash
# one-off PMC probe of the bf16x3 forward (eval): what is it waiting on?   usage (through gpurun): bash tools/pmc_b16_probe.sh
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run this through gpurun}"
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > "$O/avail.txt" 2>&1 || true
set +e
rm -rf "$O"/p_*
B="--no-cpu-baseline --no-alt --steps 3 --warmup 1 --mode ${PB_MODE:-eval}"
i=0
for c in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM" \
         "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
         "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum" ; do
  for p in ${PB_PRECS:-bf16x3 fp32}; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$O/p_${p}_$i" -o pmc -- python3 "$R/bench.py" $B --precision $p > "$O/p_${p}_$i.log" 2>&1
  done
  i=$((i+1))
done
python3 - "$O" <<'PY'
import csv, glob, os, sys
O = sys.argv[1]
for d in sorted(glob.glob(os.path.join(O, "p_*_*"))):
    if not os.path.isdir(d): continue
    f = glob.glob(os.path.join(d, "**", "pmc_counter_collection.csv"), recursive=True)
    if not f: print(d, "no csv"); continue
    acc = {}
    for r in csv.DictReader(open(f[0])):
        if "fused_fwd" not in r["Kernel_Name"]: continue
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    print(os.path.basename(d), {k: round(sum(v) / len(v)) for k, v in acc.items()})
PY
