cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_wavtok_gpu_parity.py -x -q 2>&1 | tail -4
python bench.py --codec wavtokenizer --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_r2c_wavtok.json 2> gpurun_out/bench_r2c_wavtok.err; echo "bench rc $?"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r2c -o r2c -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_r2c_prof.json 2> gpurun_out/bench_r2c_prof.err; echo "encodec prof rc $?"
tail -5 gpurun_out/bench_r2c_prof.err
