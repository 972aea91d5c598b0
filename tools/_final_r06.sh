set -uo pipefail
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"; cd "$R"
bash tools/bench_round.sh r06 > $O/bench_round_r06.log 2>&1; tail -14 $O/bench_round_r06.log
bash tools/profile_round.sh > $O/profile_round_r06.log 2>&1; tail -3 $O/profile_round_r06.log
