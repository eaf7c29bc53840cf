cd $GRAFT_REPO_ROOT
python -m pytest tests/test_dac_gpu_parity.py tests/test_dac_gpu_fullsize.py -x -q 2>&1 | grep -E "passed|failed|PARITY"
AC_PROF_DETAIL=1 python bench.py --codec dac --batch 256 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_r2o_dac.json 2>/dev/null; echo "rc $?"
