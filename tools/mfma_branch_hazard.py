"""Scan hipcc -S output (or llvm-objdump -d text) for an MFMA whose result is read by a vector instruction a few instructions
later ACROSS a taken branch: round 4 found hipcc's hazard recogniser leaving `v_mfma ... a[0:3]` -> `s_cbranch` -> `v_mov` ->
`v_accvgpr_read a3` without wait states (rvq16.h, the first WS = 4 loop: run-to-run different tokens).  Heuristic, text-level:
for every branch within LOOKBACK instructions behind an MFMA, follow the branch target (and the fall-through) and report vector
reads of the MFMA's destination registers within (matrix-pipe passes + 1) issue slots, counting s_nop N as N + 1 slots, an MFMA as its passes and every other
instruction as one.  Usage: mfma_branch_hazard.py file.s [...]"""
import re, sys

LOOKBACK = 3      # instructions between the MFMA and the branch


def passes(op):   # matrix-pipe passes of 4 cycles (lower bounds): the result is ready ~passes + 2 issue slots behind the MFMA
    if "32x32" in op: return 8
    if "16x16x4" in op or "16x16x4f32" in op: return 8
    return 4


def regs(tok):
    m = re.match(r"([av])\[(\d+):(\d+)\]", tok)
    if m: return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.match(r"([av])(\d+)$", tok)
    return {(m.group(1), int(m.group(2)))} if m else set()


def scan(path):
    lines = [l.split(";")[0].split("//")[0].rstrip() for l in open(path)]
    label_at, instrs, kernel = {}, [], None
    for l in lines:
        s = l.strip()
        if not s or s.startswith("."):
            if re.match(r"\.?L?BB\d+_\d+:", s): label_at[s[:-1]] = len(instrs)
            continue
        if s.endswith(":"):
            if s.startswith("_Z") or s.startswith("<"): kernel = s[:-1]
            label_at[s[:-1]] = len(instrs)
            continue
        instrs.append((s, kernel))
    found, seen = [], set()
    for i, (s, k) in enumerate(instrs):
        if not s.startswith("v_mfma"): continue
        dst = regs(s.split()[1].rstrip(","))
        NEED = passes(s.split()[0]) + 1                    # (what hipcc itself leaves inside a basic block)
        for j in range(i + 1, min(i + 1 + LOOKBACK, len(instrs))):
            t = instrs[j][0]
            if not t.startswith(("s_cbranch", "s_branch")): continue
            target = t.split()[-1]
            starts = [label_at.get(target)]
            if t.startswith("s_cbranch"): starts.append(j + 1)
            for st in starts:
                if st is None: continue
                slots = sum(passes(instrs[q2][0].split()[0]) if instrs[q2][0].startswith("v_mfma") else 1 for q2 in range(i + 1, j))
                q, steps = st, 0
                while q < len(instrs) and steps < 12:
                    u = instrs[q][0]
                    q += 1; steps += 1
                    if u.startswith("s_branch"):                       # follow an unconditional branch (a taken branch costs more than a slot: lower bound)
                        q = label_at.get(u.split()[-1], len(instrs)); slots += 1; continue
                    if u.startswith(("s_endpgm", "s_setpc")): break
                    if u.startswith("s_nop"): slots += int(u.split()[1]) + 1; continue
                    if slots >= NEED: break
                    ops = u.replace(",", " ").split()
                    if u.startswith("v_mfma") and len(ops) > 4 and regs(ops[4]) & dst and not any(regs(o) & dst for o in ops[2:4]):
                        break                                             # accumulation (srcC = the result): forwarded in hardware
                    if u.startswith("v_") and len(ops) > 2 and any(regs(o.lstrip("-|").rstrip("|")) & dst for o in ops[2:]):
                        if (i, q) not in seen:
                            seen.add((i, q))
                            found.append((k, s, t, u, slots))
                        break
                    if u.startswith("v_mfma") and any(regs(o) & dst for o in ops[1:2]): break     # overwritten by the next MFMA chain
                    slots += passes(ops[0]) if u.startswith("v_mfma") else 1
    return found


if __name__ == "__main__":
    bad = 0
    for p in sys.argv[1:]:
        for k, s, t, u, slots in scan(p):
            bad += 1
            print(f"{p}: {k}\n    {s}\n    {t}\n    {u}    ({slots} slots behind the MFMA)")
    print(f"{bad} suspicious MFMA -> branch -> read sequences")
    sys.exit(1 if bad else 0)
