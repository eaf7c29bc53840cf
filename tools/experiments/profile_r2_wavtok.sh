cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_wavtok_gpu_fullsize.py -x -q 2>&1 | tail -5
AC_PROF_DETAIL=1 python bench.py --codec wavtokenizer --steps 10 --warmup 3 > gpurun_out/bench_r2b_wavtok.json 2> gpurun_out/bench_r2b_wavtok.err; echo "bench rc $?"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r2b_wt -o r2b_wt -- python3 bench.py --codec wavtokenizer --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_r2b_wavtok_prof.json 2> gpurun_out/bench_r2b_wavtok_prof.err; echo "prof rc $?"
python tools/rocpd_stats.py $(ls gpurun_out/prof_r2b_wt/*/*results.db | head -1) > gpurun_out/r2b_wavtok_kernel_stats.txt; head -40 gpurun_out/r2b_wavtok_kernel_stats.txt
