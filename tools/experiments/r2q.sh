cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q 2>&1 | grep -E "passed|failed|PARITY|Error|assert"
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_r2q.json 2>/dev/null; echo "rc $?"
AC_LSTM_DBG=8 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity > gpurun_out/bench_r2q_nopf.json 2>/dev/null; echo "rc $?"
