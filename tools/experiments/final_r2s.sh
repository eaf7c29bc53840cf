cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|PARITY|Error" > gpurun_out/r2s_pytest.log; cat gpurun_out/r2s_pytest.log
cp parity_report.json gpurun_out/r2s_parity_report.json 2>/dev/null
python bench.py > gpurun_out/bench_r2s.json 2> gpurun_out/bench_r2s.err; echo "bench rc $?"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r2s -o r2s -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity > gpurun_out/bench_r2s_under_rocprof.json 2> gpurun_out/bench_r2s_prof.err; echo "prof rc $? (139 = rocprofv3's own exit crash after cooperative launches; outputs complete)"
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_r2s_fetch -o f -f csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity > /dev/null 2>&1; echo "fetch rc $?"
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_r2s_write -o w -f csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity > /dev/null 2>&1; echo "write rc $?"
python bench.py --codec wavtokenizer > gpurun_out/bench_r2s_wavtokenizer.json 2> /dev/null; echo "wavtok rc $?"
python bench.py --codec mimi --batch 128 --steps 5 --warmup 2 > gpurun_out/bench_r2s_mimi.json 2> /dev/null; echo "mimi rc $?"
python bench.py --codec dac --batch 256 --steps 2 --warmup 1 > gpurun_out/bench_r2s_dac.json 2> /dev/null; echo "dac rc $?"
python bench.py --precision bf16 > gpurun_out/bench_r2s_encodec_bf16.json 2> /dev/null; echo "bf16 rc $?"
python bench.py --precision fp32_bf16x3 --steps 10 > gpurun_out/bench_r2s_encodec_bf16x3.json 2> /dev/null; echo "bf16x3 rc $?"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r2s_wt -o r2s_wt -- python3 bench.py --codec wavtokenizer --steps 10 --warmup 3 --no-cpu-baseline --no-parity > /dev/null 2> /dev/null; echo "wt prof rc $?"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python tools/rocpd_stats.py $(find gpurun_out/prof_r2s -name "*.db" | head -1) > gpurun_out/r2s_kernel_stats.txt 2>&1; head -20 gpurun_out/r2s_kernel_stats.txt
python tools/rocpd_stats.py $(find gpurun_out/prof_r2s_wt -name "*.db" | head -1) > gpurun_out/r2s_wavtok_kernel_stats.txt 2>&1
python tools/collect_traffic.py $(find gpurun_out/pmc_r2s_fetch -name "*counter_collection.csv" | head -1) $(find gpurun_out/pmc_r2s_write -name "*counter_collection.csv" | head -1) > gpurun_out/r2s_traffic.json 2> gpurun_out/r2s_traffic.err; head -c 600 gpurun_out/r2s_traffic.json
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r2s_mimi -o r2s_mimi -- python3 bench.py --codec mimi --batch 128 --steps 3 --warmup 1 --no-cpu-baseline --no-parity > /dev/null 2> /dev/null; echo "mimi prof rc $?"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r2s_dac -o r2s_dac -- python3 bench.py --codec dac --batch 256 --steps 1 --warmup 1 --no-cpu-baseline --no-parity > /dev/null 2> /dev/null; echo "dac prof rc $?"
python tools/rocpd_stats.py $(find gpurun_out/prof_r2s_mimi -name "*.db" | head -1) > gpurun_out/r2s_mimi_kernel_stats.txt 2>&1
python tools/rocpd_stats.py $(find gpurun_out/prof_r2s_dac -name "*.db" | head -1) > gpurun_out/r2s_dac_kernel_stats.txt 2>&1
