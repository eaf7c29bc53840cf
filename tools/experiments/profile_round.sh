cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r1q -o r1q -- python3 bench.py --steps 20 --warmup 3 > gpurun_out/bench_r1q_prof.json 2> gpurun_out/bench_r1q_prof.err; echo "prof rc $?"
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_r1q_fetch -o f -f csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1; echo "fetch rc $?"
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_r1q_write -o w -f csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1; echo "write rc $?"
python bench.py > gpurun_out/bench_r1q.json 2> gpurun_out/bench_r1q.err; echo "bench rc $?"
python bench.py --codec mimi --batch 128 --steps 5 --warmup 2 > gpurun_out/bench_r1q_mimi.json 2> gpurun_out/bench_r1q_mimi.err; echo "mimi rc $?"
python bench.py --codec dac --batch 256 --steps 2 --warmup 1 > gpurun_out/bench_r1q_dac.json 2> gpurun_out/bench_r1q_dac.err; echo "dac rc $?"
ls gpurun_out/prof_r1q gpurun_out/pmc_r1q_fetch gpurun_out/pmc_r1q_write
