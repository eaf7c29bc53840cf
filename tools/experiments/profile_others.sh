cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r1r_mimi -o r1r_mimi -- python3 bench.py --codec mimi --batch 128 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_r1r_mimi_prof.json 2> /dev/null; echo "mimi rc $?"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r1r_dac -o r1r_dac -- python3 bench.py --codec dac --batch 256 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_r1r_dac_prof.json 2> /dev/null; echo "dac rc $?"
ls gpurun_out/prof_r1r_mimi gpurun_out/prof_r1r_dac
