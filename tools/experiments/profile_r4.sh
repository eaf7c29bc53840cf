# Round-4 evidence run (one gpurun call): headline bench, rocprofv3 kernel trace + stats of the same command, FETCH_SIZE / WRITE_SIZE
# passes (traffic table), SQ counter passes (matrix pipe busy, waits, instruction mix, LDS conflicts), kernel traces of the other three
# codecs.  Outputs under gpurun_out/<tag>/; the summaries are copied into profiles/ by hand.  Usage: profile_r4.sh <tag> [full]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
T=${1:-r4}
O=gpurun_out/$T
mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
rocprofv3 --kernel-trace --stats -d $O/prof -o $T -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity > $O/bench_under_rocprof.json 2> $O/prof.err; echo "prof rc $? (139 = rocprofv3's own exit crash after cooperative launches; outputs complete)"
python tools/rocpd_stats.py $(find $O/prof -name "*.db" | head -1) > $O/kernel_stats.txt 2>&1; head -18 $O/kernel_stats.txt
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o f -f csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity > /dev/null 2>&1; echo "fetch rc $?"
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o w -f csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity > /dev/null 2>&1; echo "write rc $?"
python tools/collect_traffic.py $(find $O/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $O/pmc_write -name "*counter_collection.csv" | head -1) > $O/traffic.json 2> $O/traffic.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES -d $O/pmc_sq -o sq -f csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity > /dev/null 2>&1; echo "sq rc $?"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS -d $O/pmc_sq2 -o sq2 -f csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity > /dev/null 2>&1; echo "sq2 rc $?"
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT -d $O/pmc_grbm -o g -f csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity > /dev/null 2>&1; echo "grbm rc $?"
python - "$O" > $O/sq_counters.txt <<'PY'
import csv, glob, collections, sys
O = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for d in ("pmc_sq", "pmc_sq2", "pmc_grbm"):
    fs = glob.glob(f"{O}/{d}/**/*counter_collection.csv", recursive=True)
    if not fs: print(d, "no csv"); continue
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ac::", "")[:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    if v.get("SQ_WAVE_CYCLES", 0) < 1e7: continue
    wc = v["SQ_WAVE_CYCLES"]
    print(f"{k}: launches {len(cnt[k]) // 3 or len(cnt[k])}  WAVE_CYCLES {wc:.3g}  BUSY_CYCLES {v.get('SQ_BUSY_CYCLES', 0):.3g}  active {100 * v.get('SQ_ACTIVE_INST_ANY', 0) / wc:.0f}%  wait_any {100 * v.get('SQ_WAIT_ANY', 0) / wc:.0f}%  "
          f"wait_inst {100 * v.get('SQ_WAIT_INST_ANY', 0) / wc:.0f}%  MFMA_BUSY_CYCLES {v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0):.3g}  GRBM_GUI_ACTIVE {v.get('GRBM_GUI_ACTIVE', 0):.3g}  "
          f"insts valu {v.get('SQ_INSTS_VALU', 0):.3g} mfma {v.get('SQ_INSTS_MFMA', 0):.3g} lds {v.get('SQ_INSTS_LDS', 0):.3g} salu {v.get('SQ_INSTS_SALU', 0):.3g} trans {v.get('SQ_INSTS_VALU_TRANS', 0):.3g}  "
          f"lds conflict/active {v.get('SQ_LDS_BANK_CONFLICT', 0):.3g}/{v.get('SQ_LDS_IDX_ACTIVE', 0):.3g}")
PY
cat $O/sq_counters.txt | cut -c1-400
if [ "$2" = "full" ]; then
  python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|PARITY|Error" > $O/pytest.log; cat $O/pytest.log
  cp parity_report.json $O/parity_report.json 2>/dev/null
  for c in wavtokenizer:64:10 mimi:128:3 dac:256:1; do
    IFS=: read n b s <<< "$c"
    python bench.py --codec $n --batch $b --steps $s --warmup 1 --no-cpu-baseline > $O/bench_$n.json 2> /dev/null; echo "$n rc $?"
    rocprofv3 --kernel-trace --stats -d $O/prof_$n -o ${T}_$n -- python3 bench.py --codec $n --batch $b --steps $s --warmup 1 --no-cpu-baseline --no-parity > /dev/null 2> /dev/null; echo "$n prof rc $?"
    python tools/rocpd_stats.py $(find $O/prof_$n -name "*.db" | head -1) > $O/${n}_kernel_stats.txt 2>&1
  done
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
fi
find $O -name "*.db" -size +20M -delete; find $O -name "*.csv" -size +20M -delete; du -sh $O
