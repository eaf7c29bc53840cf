cd $GRAFT_REPO_ROOT
for v in NOSPLIT NOB NOMFMA; do
AUDIOCODECS_AMD_LIB=$PWD/scratch_probe/lib_$v.so AC_PROF_DETAIL=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity > gpurun_out/bench_r2j_$v.json 2>/dev/null; echo "$v rc $?"
done
