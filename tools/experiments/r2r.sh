cd $GRAFT_REPO_ROOT
python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-parity > gpurun_out/bench_r2r_base.json 2>/dev/null
AUDIOCODECS_AMD_LIB=$PWD/scratch_probe/lib_noslp.so python bench.py --steps 15 --warmup 3 --no-cpu-baseline > gpurun_out/bench_r2r_noslp.json 2>/dev/null
python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-parity > gpurun_out/bench_r2r_base2.json 2>/dev/null
AUDIOCODECS_AMD_LIB=$PWD/scratch_probe/lib_noslp.so python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-parity > gpurun_out/bench_r2r_noslp2.json 2>/dev/null
