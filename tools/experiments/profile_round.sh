cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r1r -o r1r -- python3 bench.py --steps 20 --warmup 3 > gpurun_out/bench_r1r_prof.json 2> gpurun_out/bench_r1r_prof.err; echo "prof rc $?"
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_r1r_fetch -o f -f csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1; echo "fetch rc $?"
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_r1r_write -o w -f csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1; echo "write rc $?"
python bench.py > gpurun_out/bench_r1r.json 2> gpurun_out/bench_r1r.err; echo "bench rc $?"
python bench.py --codec mimi --batch 128 --steps 5 --warmup 2 > gpurun_out/bench_r1r_mimi.json 2> gpurun_out/bench_r1r_mimi.err; echo "mimi rc $?"
python bench.py --codec dac --batch 256 --steps 2 --warmup 1 > gpurun_out/bench_r1r_dac.json 2> gpurun_out/bench_r1r_dac.err; echo "dac rc $?"
ls gpurun_out/prof_r1r gpurun_out/pmc_r1r_fetch gpurun_out/pmc_r1r_write
