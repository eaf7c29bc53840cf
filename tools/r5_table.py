"""DESIGN.md section 5 table from the committed evidence of a round: kernel trace summary, traffic summary, SQ counters.
Usage: r5_table.py profiles/r5m_encodec_kernel_stats.txt profiles/r5m_traffic.json profiles/r5m_encodec_sq_counters.txt steps"""
import json, re, sys
stats, traffic, sq, steps = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
tr = json.load(open(traffic))["kernels"]
def norm(n): return re.sub(r"\(.*$", "", n.replace("void ", "").replace("ac::", "")).strip()
busy = {}
for line in open(sq):
    m = re.match(r"(.+?): launches (\d+).*MFMA_BUSY_CYCLES (\S+)\s+GRBM_GUI_ACTIVE (\S+)\s+insts valu (\S+) mfma (\S+)", line)
    if m and float(m.group(4)) > 0:
        busy[m.group(1).strip()] = (100 * float(m.group(3)) / (float(m.group(4)) / 8 * 1024), float(m.group(5)) / max(float(m.group(6)), 1))
print("| Kernel | launches | ms / step | rocprofv3 avg µs | measured HBM bytes / launch | matrix pipe busy; VALU per MFMA |")
print("|---|---|---|---|---|---|")
for line in open(stats):
    m = re.match(r"(.+?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", line)
    if not m: continue
    name, calls, tot, avg = m.group(1).strip(), int(m.group(2)), float(m.group(3)), float(m.group(4))
    if tot / steps < 0.1: continue
    key = norm(name).rstrip(".")
    t = next((v for k, v in tr.items() if k.startswith(key[:60]) and v.get("hbm_bytes_per_launch")), None)
    b = next((v for k, v in busy.items() if k.startswith(key[:60])), None)
    print(f"| `{key}` | {calls / steps:g} | {tot / steps:.2f} | {avg:.0f} | {t['hbm_bytes_per_launch'] / 1e9:.2f} GB |" if t else f"| `{key}` | {calls / steps:g} | {tot / steps:.2f} | {avg:.0f} | — |", end="")
    print(f" {b[0]:.1f} %; {b[1]:.1f} |" if b else " — |")
