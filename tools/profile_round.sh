#!/bin/bash
# rocprofv3 passes of one round (run on the GPU box through gpurun).  usage: tools/profile_round.sh [extra bench.py args]
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run this through gpurun (GRAFT_REPO_ROOT is the root of the repo copy on the GPU box)}"
R="$GRAFT_REPO_ROOT"
O="$R/gpurun_out"
[ -f "$R/bench.py" ] || { echo "no bench.py under $R" >&2; exit 1; }
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rm -rf "$O"/i_*
set +e          # a failing pass must not hide the others: each pass keeps its own log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/i_train -o train -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-alt > $O/i_train.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/i_eval -o eval -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-alt --mode eval > $O/i_eval.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/i_train_b16 -o train -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-alt --precision bf16x3 > $O/i_train_b16.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/i_pmc_mfma -o pmc -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt > $O/i_pmc_mfma.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/i_pmc_fetch -o pmc -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt > $O/i_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/i_pmc_write -o pmc -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt > $O/i_pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/i_pmc_mfma_eval -o pmc -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt --mode eval > $O/i_pmc_mfma_eval.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/i_pmc_fetch_eval -o pmc -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt --mode eval > $O/i_pmc_fetch_eval.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/i_pmc_write_eval -o pmc -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt --mode eval > $O/i_pmc_write_eval.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/i_pmc_b16 -o pmc -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt --precision bf16x3 > $O/i_pmc_b16.log 2>&1
find "$O"/i_* -name "*kernel_trace.csv" -size +1M -delete
ls -R "$O"/i_* | head -50
