cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
rm -rf $O/i_*
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
find $O/i_* -name "*kernel_trace.csv" -size +1M -delete
ls -R $O/i_* | head -50
