cd $GRAFT_REPO_ROOT
python -m pytest tests/test_bf16_mode_gpu.py tests/test_gpu_parity.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | head -5
python - <<'PY'
import json
d=json.load(open('gpurun_out/parity_report.json'))
for k,v in d['cases'].items():
    if 'bf16' in k: print(k, v)
PY
for c in encodec wavtokenizer; do
python bench.py --codec $c --precision bf16 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_r2s_${c}_bf16.json 2>/dev/null; echo "$c rc $?"
done
python bench.py --codec mimi --batch 128 --precision bf16 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_r2s_mimi_bf16.json 2>/dev/null
