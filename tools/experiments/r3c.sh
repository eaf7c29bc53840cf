cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_fused_chains_gpu.py -q 2>&1 | tail -4
python bench.py --no-cpu-baseline --no-other-configs --no-exact > gpurun_out/r3c_bench.json 2> gpurun_out/r3c_bench.err; echo "bench rc $?"
python - <<'PY'
import json
for f in ("r3c_bench",):
    try:
        d = json.load(open(f"gpurun_out/{f}.json"))
        print(f, d["ms_per_step"], d["value"], d.get("parity"))
        for k in d["kernels"]: print("   ", k["name"], k["launches_per_step"], k["ms_per_step"], k["tflops"], k["gbs"])
    except Exception as e: print(f, "ERR", e)
PY
