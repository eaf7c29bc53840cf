cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|PARITY|Error|FAILED" > gpurun_out/r3m_pytest.log; cat gpurun_out/r3m_pytest.log
python bench.py > gpurun_out/r3m_bench.json 2> gpurun_out/r3m_bench.err; echo "bench rc $?"; tail -2 gpurun_out/r3m_bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3m_bench.json"))
print(d["ms_per_step"], d["value"], d["parity"], d.get("exact_fp32_ms_per_step"), d["roofline"]["frac"], d["whole_path"])
for k, v in d.get("other_configs", {}).items(): print(k, {kk: v.get(kk) for kk in ("value", "ms_per_step", "error")}, (v.get("parity") or {}).get("token_exact_match"))
print(d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
