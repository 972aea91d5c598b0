for i in 1 2; do
for v in base new; do
  if [ $v = base ]; then export CFNERF_LIB=$GRAFT_REPO_ROOT/cf-nerf_amd/build/ab/libcfnerf_base.so; else unset CFNERF_LIB; fi
  for m in "" "--precision bf16x3"; do
    python bench.py --steps 30 --no-cpu-baseline --no-alt $m 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v','train','$m',round(d['value']),round(d['ms_per_step'],3),d['kernel_ms']['fwd'],d['kernel_ms']['bwd_data'])"
    python bench.py --steps 30 --no-cpu-baseline --no-alt --mode eval $m 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v','eval','$m',round(d['value']),round(d['ms_per_step'],3))"
  done
done
done
