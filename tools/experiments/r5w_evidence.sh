# Round-5 closing evidence (one gpurun call): the GPU suite, the default bench line, and the rocprofv3 kernel trace + stats of the bench command of each codec.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
T=${1:-r5w}
O=gpurun_out/$T
mkdir -p $O
python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc $?"; grep -E "passed|failed" $O/pytest.log | tail -2
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; cut -c1-400 $O/bench.json
cp gpurun_out/parity_report.json $O/parity_report.json 2>/dev/null
for n in encodec wavtokenizer mimi dac; do
  case $n in encodec) b=64;; wavtokenizer) b=64;; mimi) b=128;; dac) b=39;; esac
  rocprofv3 --kernel-trace --stats -d $O/prof_$n -o ${T}_$n -- python3 bench.py --codec $n --batch $b --no-cpu-baseline --no-parity --steps 10 --warmup 2 > $O/bench_${n}_under_rocprof.json 2> $O/prof_$n.err; echo "$n prof rc $?"
  python tools/rocpd_stats.py $(find $O/prof_$n -name "*.db" | head -1) > $O/${n}_kernel_stats.txt 2>&1; head -8 $O/${n}_kernel_stats.txt | cut -c1-150
done
find $O -name "*.db" -size +20M -delete; find $O -name "*.csv" -size +20M -delete; du -sh $O
