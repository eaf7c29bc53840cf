cd $GRAFT_REPO_ROOT
python -m pytest tests/test_dac_gpu_parity.py -x -q 2>&1 | grep -E "passed|failed|FAILED|Error|PARITY" | head
for e in 1 0; do
AC_DAC_PAD=$e python bench.py --codec dac --batch 256 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r3n_dac_$e.json 2>/dev/null
python - <<PY
import json
d = json.load(open("gpurun_out/r3n_dac_$e.json"))
print("pad=$e", d["ms_per_step"], d["value"], d["parity"]["token_exact_match"], d["parity"]["decode_rms_err"])
for k in d["kernels"][:7]: print(f'   {k["ms_per_step"]:9.2f} ms {k["tflops"]:7.1f} TF {k["gbs"]:7.0f} GB/s x{k["launches_per_step"]:.0f} {k["name"]}')
PY
done
