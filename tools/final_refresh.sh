set -uo pipefail
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"; cd "$R"
python -m pytest tests -m gpu -q -x > $O/r4_t13.log 2>&1; echo "pytest rc=$?" >> $O/r4_t13.log; tail -3 $O/r4_t13.log
bash tools/bench_round.sh r04 > $O/bench_round_r04.log 2>&1; tail -12 $O/bench_round_r04.log
bash tools/profile_stats_only.sh > $O/profile_stats_r04.log 2>&1; tail -4 $O/profile_stats_r04.log
