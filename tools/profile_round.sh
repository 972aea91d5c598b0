#!/bin/bash
# rocprofv3 passes of one round (run on the GPU box through gpurun):  tools/profile_round.sh
#   kernel-trace + stats per workload, then counters in their OWN passes (never combined with a trace domain other
#   than --kernel-trace): MFMA-busy / wave-cycle counters, FETCH_SIZE, WRITE_SIZE.
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run this through gpurun (GRAFT_REPO_ROOT is the root of the repo copy on the GPU box)}"
R="$GRAFT_REPO_ROOT"
O="$R/gpurun_out"
[ -f "$R/bench.py" ] || { echo "no bench.py under $R" >&2; exit 1; }
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rm -rf "$O"/i_*
set +e          # a failing pass must not hide the others: each pass keeps its own log
B="--no-cpu-baseline --no-alt"
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY"
stats() {  # name, bench args
  local n="$1"; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/i_$n" -o "$n" -- python3 "$R/bench.py" --steps 50 --warmup 3 $B "$@" > "$O/i_$n.log" 2>&1
}
pmc() {    # name, counters, bench args
  local n="$1" c="$2"; shift 2
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$O/i_$n" -o pmc -- python3 "$R/bench.py" --steps 3 --warmup 1 $B "$@" > "$O/i_$n.log" 2>&1
}
stats train
stats eval --mode eval
stats train_b16 --precision bf16x3
stats w512 --config W512
stats c4 --config C4
stats c3 --config C3
stats k64 --config K64
pmc pmc_mfma "$SQ"
pmc pmc_fetch FETCH_SIZE
pmc pmc_write WRITE_SIZE
pmc pmc_mfma_eval "$SQ" --mode eval
pmc pmc_fetch_eval FETCH_SIZE --mode eval
pmc pmc_write_eval WRITE_SIZE --mode eval
pmc pmc_b16 "$SQ" --precision bf16x3
pmc pmc_mfma_w512 "$SQ" --config W512
pmc pmc_fetch_w512 FETCH_SIZE --config W512
pmc pmc_write_w512 WRITE_SIZE --config W512
find "$O"/i_* -name "*kernel_trace.csv" -size +1M -delete
ls -R "$O"/i_* | head -80
