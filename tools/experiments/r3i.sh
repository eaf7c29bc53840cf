cd $GRAFT_REPO_ROOT
AC_PROF_DETAIL=1 python bench.py --no-cpu-baseline --no-other-configs --no-exact --no-parity > gpurun_out/r3i_detail.json 2>/dev/null
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3i_detail.json"))
print(d["ms_per_step"])
for k in d["kernels"]: print(f'{k["ms_per_step"]:7.3f} ms  {k["tflops"]:7.1f} TF  {k["gbs"]:7.0f} GB/s  x{k["launches_per_step"]:.0f}  {k["name"]}')
PY
