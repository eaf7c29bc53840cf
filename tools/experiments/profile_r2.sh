cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|PARITY|Error" > gpurun_out/r2p_pytest.log; cat gpurun_out/r2p_pytest.log
python bench.py > gpurun_out/bench_r2p.json 2> gpurun_out/bench_r2p.err; echo "bench rc $?"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r2p -o r2p -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity > gpurun_out/bench_r2p_prof.json 2> gpurun_out/bench_r2p_prof.err; echo "prof rc $? (139 = rocprofv3's own exit crash after cooperative launches; outputs complete)"
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_r2p_fetch -o f -f csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity > /dev/null 2>&1; echo "fetch rc $?"
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_r2p_write -o w -f csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity > /dev/null 2>&1; echo "write rc $?"
python bench.py --codec wavtokenizer > gpurun_out/bench_r2p_wavtok.json 2> gpurun_out/bench_r2p_wavtok.err; echo "wavtok rc $?"
python bench.py --codec mimi --batch 128 --steps 5 --warmup 2 > gpurun_out/bench_r2p_mimi.json 2> gpurun_out/bench_r2p_mimi.err; echo "mimi rc $?"
python bench.py --codec dac --batch 256 --steps 2 --warmup 1 > gpurun_out/bench_r2p_dac.json 2> gpurun_out/bench_r2p_dac.err; echo "dac rc $?"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r2p_wt -o r2p_wt -- python3 bench.py --codec wavtokenizer --steps 10 --warmup 3 --no-cpu-baseline --no-parity > gpurun_out/bench_r2p_wavtok_prof.json 2> /dev/null; echo "wt prof rc $?"
find gpurun_out/prof_r2p gpurun_out/pmc_r2p_fetch gpurun_out/pmc_r2p_write gpurun_out/prof_r2p_wt -type f | head -20
