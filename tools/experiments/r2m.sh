cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_r2n.json 2>/dev/null; echo "rc $?"
python bench.py --codec wavtokenizer --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_r2n_wt.json 2>/dev/null
python bench.py --codec mimi --batch 128 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_r2n_mimi.json 2>/dev/null
python bench.py --codec dac --batch 256 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_r2n_dac.json 2>/dev/null
