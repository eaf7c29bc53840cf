cd $GRAFT_REPO_ROOT
for v in 1 3; do
AC_PROF_DETAIL=1 AC_T6_VARIANT=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_r2h_v$v.json 2>/dev/null; echo "v$v rc $?"
AC_T6_VARIANT=$v python bench.py --codec wavtokenizer --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_r2h_wt_v$v.json 2>/dev/null
done
