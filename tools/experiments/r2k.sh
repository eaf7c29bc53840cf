cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_wavtok_gpu_parity.py tests/test_mimi_gpu_parity.py tests/test_dac_gpu_parity.py -x -q 2>&1 | tail -4
AC_PROF_DETAIL=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_r2k.json 2>/dev/null; echo "rc $?"
python bench.py --codec wavtokenizer --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_r2k_wt.json 2>/dev/null
python bench.py --codec mimi --batch 128 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_r2k_mimi.json 2>/dev/null
