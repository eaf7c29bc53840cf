# usage: ab_env.sh VAR  -> EnCodec bench with VAR=0 and VAR=1 (per-shape detail)
for v in 0 1; do
  env $1=$v AC_PROF_DETAIL=1 python bench.py --no-cpu-baseline --steps 8 --warmup 3 > gpurun_out/ab_$v.json 2>/dev/null
done
python - <<PY
import json
a=json.load(open("gpurun_out/ab_0.json")); b=json.load(open("gpurun_out/ab_1.json"))
print("$1=0:", a["ms_per_step"], " $1=1:", b["ms_per_step"])
ka={k["name"].split("> ")[-1] if "tap_gemm6" in k["name"] else k["name"]:k for k in a["kernels"]}
for k in b["kernels"]:
    n=k["name"].split("> ")[-1] if "tap_gemm6" in k["name"] else k["name"]
    if n in ka and abs(k["ms_per_step"]-ka[n]["ms_per_step"])>0.02: print(f"  {n:40s} {ka[n]['ms_per_step']:.3f} -> {k['ms_per_step']:.3f}  ({k['name'][:28]})")
PY
