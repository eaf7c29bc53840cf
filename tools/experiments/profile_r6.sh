# Round-6 counter passes for ONE codec (one gpurun call): rocprofv3 kernel trace + stats, FETCH_SIZE / WRITE_SIZE (traffic table, gfx950
# FETCH x2 correction in tools/collect_traffic.py), two SQ passes + GRBM.  Counter passes never combine --pmc with a trace domain.
# Usage: profile_r6.sh <tag> <codec> [trace|traffic|sq ...]   (default: all three)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
T=${1:-r6}; n=${2:-encodec}; shift; shift
WHAT=${@:-"trace traffic sq"}
O=gpurun_out/$T
mkdir -p $O
case $n in encodec) b=64;; wavtokenizer) b=64;; mimi) b=128;; dac) b=39;; esac
A="--codec $n --batch $b --no-cpu-baseline --no-parity"
for w in $WHAT; do
case $w in
trace)
  rocprofv3 --kernel-trace --stats -d $O/prof_$n -o ${T}_$n -- python3 bench.py $A --steps 10 --warmup 2 > $O/bench_${n}_under_rocprof.json 2> $O/prof_$n.err; echo "$n prof rc $?"
  python tools/rocpd_stats.py $(find $O/prof_$n -name "*.db" | head -1) > $O/${n}_kernel_stats.txt 2>&1; head -14 $O/${n}_kernel_stats.txt;;
traffic)
  rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch_$n -o f -f csv -- python3 bench.py $A --steps 1 --warmup 1 > /dev/null 2>&1; echo "$n fetch rc $?"
  rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write_$n -o w -f csv -- python3 bench.py $A --steps 1 --warmup 1 > /dev/null 2>&1; echo "$n write rc $?"
  python tools/collect_traffic.py $(find $O/pmc_fetch_$n -name "*counter_collection.csv" | head -1) $(find $O/pmc_write_$n -name "*counter_collection.csv" | head -1) > $O/${n}_traffic.json 2> $O/${n}_traffic.err
  python - $O/${n}_traffic.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
rows = d.get("kernels", d) if isinstance(d, dict) else d
print(json.dumps(rows)[:3000])
PY
  ;;
sq)
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES -d $O/pmc_sq_$n -o sq -f csv -- python3 bench.py $A --steps 1 --warmup 1 > /dev/null 2>&1; echo "$n sq rc $?"
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS -d $O/pmc_sq2_$n -o sq2 -f csv -- python3 bench.py $A --steps 1 --warmup 1 > /dev/null 2>&1; echo "$n sq2 rc $?"
  rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT -d $O/pmc_grbm_$n -o g -f csv -- python3 bench.py $A --steps 1 --warmup 1 > /dev/null 2>&1; echo "$n grbm rc $?"
  python - "$O" "$n" > $O/${n}_sq_counters.txt <<'PY'
import csv, glob, collections, sys
O, n = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for d in ("pmc_sq", "pmc_sq2", "pmc_grbm"):
    fs = glob.glob(f"{O}/{d}_{n}/**/*counter_collection.csv", recursive=True)
    if not fs: print(d, "no csv"); continue
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ac::", "")[:70]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    if v.get("SQ_WAVE_CYCLES", 0) < 1e7: continue
    wc = v["SQ_WAVE_CYCLES"]
    print(f"{k}: launches {len(cnt[k]) // 3 or len(cnt[k])}  WAVE_CYCLES {wc:.3g}  BUSY_CYCLES {v.get('SQ_BUSY_CYCLES', 0):.3g}  active {100 * v.get('SQ_ACTIVE_INST_ANY', 0) / wc:.0f}%  wait_any {100 * v.get('SQ_WAIT_ANY', 0) / wc:.0f}%  "
          f"wait_inst {100 * v.get('SQ_WAIT_INST_ANY', 0) / wc:.0f}%  MFMA_BUSY_CYCLES {v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0):.3g}  GRBM_GUI_ACTIVE {v.get('GRBM_GUI_ACTIVE', 0):.3g}  "
          f"insts valu {v.get('SQ_INSTS_VALU', 0):.3g} mfma {v.get('SQ_INSTS_MFMA', 0):.3g} lds {v.get('SQ_INSTS_LDS', 0):.3g} salu {v.get('SQ_INSTS_SALU', 0):.3g} trans {v.get('SQ_INSTS_VALU_TRANS', 0):.3g}  "
          f"lds conflict/active {v.get('SQ_LDS_BANK_CONFLICT', 0):.3g}/{v.get('SQ_LDS_IDX_ACTIVE', 0):.3g}")
PY
  python tools/sq_table.py $O/${n}_sq_counters.txt > $O/${n}_sq_table.md 2>/dev/null; cat $O/${n}_sq_table.md | cut -c1-260;;
esac
done
find $O -name "*.db" -size +20M -delete; find $O -name "*.csv" -size +20M -delete; du -sh $O
