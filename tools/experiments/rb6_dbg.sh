for d in 0 1 2 3 4 8 16 31 7; do
AC_RB6_DBG=$d python bench.py --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/b_dbg.json 2>/dev/null
python - <<PY
import json
d=json.load(open("gpurun_out/b_dbg.json"))
print("dbg $d", d["ms_per_step"], [(k["name"][17:25], k["ms_per_step"]) for k in d["kernels"] if "rb_fused6" in k["name"]])
PY
done
