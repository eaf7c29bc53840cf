"""profiles/r4*_sq_counters.txt (the per-kernel sums tools/experiments/profile_r4.sh prints) -> the markdown table of
profiles/r4_sq_counters.md.  Matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)."""
import re, sys
rows = []
for line in open(sys.argv[1]):
    m = re.match(r"(.+?): launches (\d+)\s+WAVE_CYCLES (\S+)\s+BUSY_CYCLES (\S+)\s+active (\d+)%\s+wait_any (\d+)%\s+wait_inst (\d+)%\s+MFMA_BUSY_CYCLES (\S+)\s+GRBM_GUI_ACTIVE (\S+)\s+"
                 r"insts valu (\S+) mfma (\S+) lds (\S+) salu (\S+) trans (\S+)\s+lds conflict/active (\S+)/(\S+)", line)
    if not m: continue
    name, n = m.group(1), int(m.group(2))
    wc, bc, act, wa, wi, mb, grbm, valu, mf, lds, salu, trans, lc, la = [float(x) for x in m.groups()[2:]]
    if mf == 0 or grbm == 0: continue
    rows.append((wc, f"| `{name}` ({n}) | **{100 * mb / (grbm / 8 * 1024):.1f} %** | {act:.0f} % | {wa:.0f} % | {wi:.0f} % | {valu / mf:.1f} | {salu / mf:.1f} | {lds / mf:.2f} | {lc / la if la else 0:.2f} |"))
print("| kernel (launches per step) | matrix pipe busy | ACTIVE_INST_ANY | WAIT_ANY (waitcnt / barrier) | WAIT_INST_ANY (issue stall) | VALU per MFMA | SALU per MFMA | LDS per MFMA | LDS_BANK_CONFLICT / LDS_IDX_ACTIVE |")
print("|---|---|---|---|---|---|---|---|---|")
for _, r in sorted(rows, reverse=True): print(r)
