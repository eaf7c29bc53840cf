# Round-3 evidence run (one gpurun call): GPU test log, headline bench, rocprofv3 kernel trace + stats of the same command,
# FETCH_SIZE / WRITE_SIZE passes for the traffic table, benches + traces of the other three codecs.  Outputs under gpurun_out/;
# the summaries are copied into profiles/ by hand (tools/experiments/README.md).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
T=${1:-r3}
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|PARITY|Error" > gpurun_out/${T}_pytest.log; cat gpurun_out/${T}_pytest.log
cp parity_report.json gpurun_out/${T}_parity_report.json 2>/dev/null
python bench.py > gpurun_out/bench_${T}.json 2> gpurun_out/bench_${T}.err; echo "bench rc $?"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${T} -o ${T} -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity > gpurun_out/bench_${T}_under_rocprof.json 2> gpurun_out/bench_${T}_prof.err; echo "prof rc $? (139 = rocprofv3's own exit crash after cooperative launches; outputs complete)"
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_${T}_fetch -o f -f csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity > /dev/null 2>&1; echo "fetch rc $?"
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_${T}_write -o w -f csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity > /dev/null 2>&1; echo "write rc $?"
python bench.py --codec wavtokenizer --no-cpu-baseline > gpurun_out/bench_${T}_wavtokenizer.json 2> /dev/null; echo "wavtok rc $?"
python bench.py --codec mimi --batch 128 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_${T}_mimi.json 2> /dev/null; echo "mimi rc $?"
python bench.py --codec dac --batch 256 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_${T}_dac.json 2> /dev/null; echo "dac rc $?"
python bench.py --precision fp32_exact --steps 5 --no-cpu-baseline > gpurun_out/bench_${T}_encodec_fp32_exact.json 2> /dev/null; echo "exact rc $?"
python bench.py --precision bf16 --no-cpu-baseline > gpurun_out/bench_${T}_encodec_bf16.json 2> /dev/null; echo "bf16 rc $?"
for c in wavtokenizer:64:10 mimi:128:3 dac:256:1; do
  IFS=: read n b s <<< "$c"
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${T}_$n -o ${T}_$n -- python3 bench.py --codec $n --batch $b --steps $s --warmup 1 --no-cpu-baseline --no-parity > /dev/null 2> /dev/null; echo "$n prof rc $?"
  python tools/rocpd_stats.py $(find gpurun_out/prof_${T}_$n -name "*.db" | head -1) > gpurun_out/${T}_${n}_kernel_stats.txt 2>&1
done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python tools/rocpd_stats.py $(find gpurun_out/prof_${T} -name "*.db" | head -1) > gpurun_out/${T}_kernel_stats.txt 2>&1; head -16 gpurun_out/${T}_kernel_stats.txt
python tools/collect_traffic.py $(find gpurun_out/pmc_${T}_fetch -name "*counter_collection.csv" | head -1) $(find gpurun_out/pmc_${T}_write -name "*counter_collection.csv" | head -1) > gpurun_out/${T}_traffic.json 2> gpurun_out/${T}_traffic.err; head -c 400 gpurun_out/${T}_traffic.json
