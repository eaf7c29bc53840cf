cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_bf16_mode_gpu.py tests/test_gpu_parity.py -x -q 2>&1 | tail -5
for c in encodec wavtokenizer; do
python bench.py --codec $c --precision bf16 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_r2e_${c}_bf16.json 2> gpurun_out/bench_r2e_${c}_bf16.err; echo "$c bf16 rc $?"
done
python bench.py --codec mimi --batch 128 --precision bf16 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_r2e_mimi_bf16.json 2> gpurun_out/bench_r2e_mimi_bf16.err; echo "mimi rc $?"
python bench.py --codec mimi --batch 128 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_r2e_mimi.json 2> gpurun_out/bench_r2e_mimi.err; echo "mimi rc $?"
cat gpurun_out/parity_report.json | python -c "import json,sys; d=json.load(sys.stdin); [print(k, v) for k,v in d['cases'].items() if 'bf16' in k]"
