cd $GRAFT_REPO_ROOT
AC_T6_VARIANT=4 python -m pytest tests/test_gpu_parity.py -x -q -k "golden or oracle" 2>&1 | tail -3
AC_T6_VARIANT=4 AC_PROF_DETAIL=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_r2k4.json 2>/dev/null; echo "rc $?"
AC_T6_VARIANT=4 python bench.py --codec wavtokenizer --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_r2k4_wt.json 2>/dev/null
