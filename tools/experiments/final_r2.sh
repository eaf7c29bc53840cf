cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|PARITY|Error" > gpurun_out/r2z_pytest.log; cat gpurun_out/r2z_pytest.log
python bench.py > gpurun_out/bench_r2z.json 2> gpurun_out/bench_r2z.err; echo "bench rc $?"
python bench.py --codec wavtokenizer > gpurun_out/bench_r2z_wavtok.json 2>/dev/null; echo "wt rc $?"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
