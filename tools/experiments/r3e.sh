cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|PARITY|Error|FAILED" > gpurun_out/r3e_pytest.log; cat gpurun_out/r3e_pytest.log
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES -d gpurun_out/pmc_r3e_sq -o sq -f csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity > /dev/null 2>&1; echo "sq rc $?"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS -d gpurun_out/pmc_r3e_sq2 -o sq2 -f csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity > /dev/null 2>&1; echo "sq2 rc $?"
python - <<'PY'
import csv, glob, collections
for d in ("pmc_r3e_sq", "pmc_r3e_sq2"):
    fs = glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True)
    if not fs: print(d, "no csv"); continue
    rows = list(csv.DictReader(open(fs[0])))
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in rows:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ac::", "")[:44]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in agg.items():
        if any(s in k for s in ("enc_front", "dec_tail", "rb_fused6", "rb128", "tap_gemm6")):
            print(d, k, {c: f"{x:.3g}" for c, x in v.items()})
PY
