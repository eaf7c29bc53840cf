cd $GRAFT_REPO_ROOT
python -m pytest tests/test_split16_gpu.py tests/test_dac_gpu_fullsize.py -x -q 2>&1 | grep -E "passed|failed|FAILED|Error|assert|PARITY" | head -20
python - <<'PY'
import json
d = json.load(open("parity_report.json"))
for c in d.get("cases", d) if isinstance(d, dict) else []:
    pass
print(json.dumps([x for x in (d["cases"] if "cases" in d else d) if "speech" in json.dumps(x)][:2])[:1500])
PY
