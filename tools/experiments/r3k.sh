cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_split16_gpu.py tests/test_shape_sweep_gpu.py -x -q -k "not mimi and not dac and not wavtok" 2>&1 | grep -E "passed|failed|FAILED|Error|PARITY" | head
python bench.py --no-cpu-baseline --no-other-configs --no-exact > gpurun_out/r3k_bench.json 2>/dev/null
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3k_bench.json"))
print(d["ms_per_step"], d["value"], d["parity"]["token_exact_match"], d["parity"]["token_mismatches_outside_fp64_near_ties"], d["parity"]["decode_rms_err"])
for k in d["kernels"]: print(f'{k["ms_per_step"]:7.3f} ms  {k["tflops"]:7.1f} TF  {k["gbs"]:7.0f} GB/s  x{k["launches_per_step"]:.0f}  {k["name"]}')
PY
