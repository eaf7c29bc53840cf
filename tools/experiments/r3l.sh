cd $GRAFT_REPO_ROOT
for e in direct staged; do
AC_TAP_EPI=$e AC_PROF_DETAIL=1 python bench.py --no-cpu-baseline --no-other-configs --no-exact > gpurun_out/r3l_$e.json 2>/dev/null
python - <<PY
import json
d = json.load(open("gpurun_out/r3l_$e.json"))
print("$e", d["ms_per_step"], d["value"], d["parity"]["token_exact_match"], d["parity"]["token_mismatches_outside_fp64_near_ties"], d["parity"]["decode_rms_err"])
for k in d["kernels"]:
    if "tap_gemm6" in k["name"]: print(f'   {k["ms_per_step"]:7.3f} ms  {k["tflops"]:7.1f} TF  {k["gbs"]:7.0f} GB/s  x{k["launches_per_step"]:.0f}  {k["name"]}')
PY
done
