"""Build-time structural check of tap_gemm8_kernel's load pipeline (csrc/tap_gemm8.h) in hipcc's device assembly.

The kernel's weight stages travel L2 -> LDS as `global_load_lds_dwordx4` requests issued from inline asm: hipcc does not know they exist,
so the ONLY thing that orders a workgroup's reads of a ring slot behind them is the kernel's own counted wait in front of the stage
barrier, `s_waitcnt vmcnt(A_SLOTS)` (t8_wait): "everything but this stage's own A_SLOTS activation loads has landed".  vmcnt retires in
issue order, so that sentence is true only if, on EVERY path into the wait, the A_SLOTS youngest vector-memory requests are the
compiler-visible activation loads (`buffer_load_dwordx4`) -- if hipcc sinks one of them into a branch only some waves take, drops one,
or moves one in front of the LDS-DMA requests, a weight fragment stays in flight across the barrier and a wave multiplies stale LDS
bytes: silent, run-to-run different results.  Exactly that happened in round 4 (profiles/r4_tapgemm8.md section 2.1: the fifth load sunk
into the one-wave branch that stores it) and passed every test.

For every instantiation of tap_gemm8_kernel in the given .s files this scan builds the control-flow graph of the kernel's instruction
list (paths that hipcc's structurised `if` flags make impossible -- both arms of one `if` skipped, or both taken -- are pruned: Walker) and
checks:
  1. every inline-asm `s_waitcnt vmcnt(N)` with N > 0 (the kernel's own waits sit between `;;#ASMSTART` / `;;#ASMEND`): walking BACKWARDS
     along every path, the first N vector-memory instructions met are `buffer_load_dwordx4` (no `... lds`, no `global_load_lds*`, no store or
     atomic) -- conditional branches are followed both ways, so a load inside a region some waves skip (`s_cbranch_execz`) fails the
     path that skips it; N must be the instantiation's A_SLOTS (from its template arguments);
  2. every wait of the kernel's own (`s_waitcnt vmcnt(..)` from inline asm) is followed directly by its `s_barrier` (only further
     `s_waitcnt`s in between): the hand-over "my requests have landed -> everybody may read" is intact;
  3. walking backwards from every `s_endpgm`, a covering wait -- the counted wait, or any `vmcnt(0)` -- is met before any LDS-DMA request:
     nothing is in flight when the staged epilogue reuses the LDS or the wave ends;
  4. the kernel contains LDS-DMA requests at all (otherwise the scan is looking at the wrong thing).
(Requests stay in flight ACROSS barriers by design -- that is the pipeline: the lock-step loop keeps them over one stage, the ping-pong
loop over three segment barriers; what must hold is that the wait covering them precedes the barrier in front of their first read.)
Exit status 1 and one line per finding if anything fails.  Usage: tap8_pipeline_scan.py file.s [...]
(tests/test_tap8_pipeline_scan.py feeds it synthetic assembly with each failure.)"""
import re
import sys

KERNEL = "tap_gemm8_kernel"
VMEM = ("buffer_load", "buffer_store", "buffer_atomic", "global_load", "global_store", "global_atomic", "flat_load", "flat_store", "flat_atomic",
        "scratch_load", "scratch_store")


def a_slots(mangled):
    """A_SLOTS of Tap8Cfg<WGM, WGN, WMT, WN> from the mangled instantiation (ILi2ELi4ELi4ELi2E...): rows = 32 WGM WMT + 7 halo rows of
    8 sixteen-byte columns, over 512 threads."""
    m = re.search(KERNEL + r"ILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E", mangled)
    if not m:
        return None
    wgm, _, wmt, _ = (int(g) for g in m.groups())
    return ((32 * wgm * wmt + 7) * 8 + 511) // 512


def is_vmem(op):
    return op.startswith(VMEM)


def is_lds_dma(text):
    op = text.split()[0]
    return op.startswith("global_load_lds") or (op.startswith("buffer_load") and re.search(r"\blds\b", text) is not None)


def parse(path):
    """{kernel symbol: [(text, in_asm)]} + {kernel: {label: index}} for the kernels of interest."""
    kernels, labels = {}, {}
    cur, in_asm = None, False
    for raw in open(path):
        line = raw.rstrip("\n")
        s = line.strip()
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        s = s.split(";")[0].strip()
        if not s:
            continue
        m = re.match(r"^([A-Za-z_.$][\w.$]*):", s)
        if m:
            name = m.group(1)
            if name.startswith("_Z") and not name.startswith(".L"):
                cur = name if KERNEL in name else None
                if cur is not None:
                    kernels[cur], labels[cur] = [], {}
            elif cur is not None:
                labels[cur][name] = len(kernels[cur])
            continue
        if s.startswith("."):
            if s.startswith((".end_amdhsa_kernel", ".section", ".Lfunc_end")) and cur is not None and s.startswith(".Lfunc_end"):
                cur = None
            continue
        if cur is not None:
            kernels[cur].append((s, in_asm))
    return kernels, labels


def sreg_set(tok):
    """Scalar registers named by an operand token: s[0:1] -> {0, 1}, s5 -> {5}, vcc -> {'vcc'}; anything else -> empty."""
    tok = tok.strip().rstrip(",")
    m = re.match(r"^s\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"^s(\d+)$", tok)
    if m:
        return {int(m.group(1))}
    if tok in ("vcc", "vcc_lo", "vcc_hi"):
        return {"vcc"}
    return set()


def branch_flag(ins, p):
    """hipcc's structurised control flow carries `if` conditions across blocks in flag registers: `s_mov_b64 s[a:b], -1 | 0` ...
    `s_and_b64 vcc, exec, s[a:b]` / `s_andn2_b64 vcc, exec, s[a:b]` ... `s_cbranch_vccz | vccnz L`.  For the conditional branch at p return
    (flag operand text, value the flag must have for the branch to be TAKEN: 0 or -1), or None when the condition is anything else."""
    t = ins[p][0]
    op = t.split()[0]
    if op not in ("s_cbranch_vccz", "s_cbranch_vccnz"):
        return None
    j = p - 1
    while j >= 0 and p - j <= 6:
        u = ins[j][0]
        f = u.replace(",", " ").split()
        if f[0] in ("s_and_b64", "s_andn2_b64") and len(f) == 4 and f[1] == "vcc" and f[2] == "exec" and sreg_set(f[3]):
            zero_when_flag = 0 if f[0] == "s_and_b64" else -1        # the flag value that makes vcc zero
            taken_flag = zero_when_flag if op == "s_cbranch_vccz" else (-1 - zero_when_flag)
            return f[3], taken_flag
        if f[0].startswith(("s_cbranch", "s_branch")) or (len(f) > 1 and "vcc" in sreg_set(f[1])) or f[0].startswith("v_cmp"):
            return None
        j -= 1
    return None


class Walker:
    """Backward walk over a kernel's control-flow graph that prunes paths the flag idiom above makes impossible (both arms of one `if`
    skipped or both taken).  visit(j, state) -> (verdict, new state): verdict None = keep walking, True = this path is fine,
    a string = finding."""

    def __init__(self, ins, lab):
        self.ins = ins
        self.target = {}
        succ = [[] for _ in ins]
        for i, (t, _) in enumerate(ins):
            op = t.split()[0]
            if op == "s_endpgm" or op.startswith("s_setpc"):
                continue
            tgt = lab.get(t.split()[-1]) if op.startswith(("s_cbranch", "s_branch")) else None
            if tgt is not None and tgt >= len(ins):
                tgt = None
            if op == "s_branch":
                if tgt is not None:
                    succ[i].append(tgt)
                continue
            if i + 1 < len(ins):
                succ[i].append(i + 1)
            if op.startswith("s_cbranch") and tgt is not None:
                succ[i].append(tgt)
                self.target[i] = tgt
        self.pred = [[] for _ in ins]
        for i, ss in enumerate(succ):
            for j in ss:
                self.pred[j].append(i)
        # only registers that ARE flags (written by `s_mov_b64 reg, 0 | -1` somewhere) can ever decide a path: conditions computed any other
        # way would only multiply the walk's states
        flagregs = set()
        for t, _ in ins:
            f = t.replace(",", " ").split()
            if f[0] == "s_mov_b64" and len(f) == 3 and f[2] in ("0", "-1"):
                flagregs.add(f[1])
        self.flags = {}
        for i in self.target:
            fl = branch_flag(ins, i)
            if fl is not None and fl[0] in flagregs:
                self.flags[i] = fl

    def walk(self, start, visit, state0, entry_verdict):
        """start: the instruction whose predecessors the walk begins at."""
        seen = set()
        stack = [(p, start, state0, frozenset()) for p in self.pred[start]]
        while stack:
            j, came_from, st, cons = stack.pop()
            # the edge j -> came_from: which way did a conditional branch at j go?
            fl = self.flags.get(j)
            if fl is not None and self.target[j] != j + 1:
                taken = came_from == self.target[j]
                want = fl[1] if taken else (-1 - fl[1])
                cons = frozenset(set(cons) | {(fl[0], want)})
            # the instruction at j may define a flag some constraint speaks about
            f = self.ins[j][0].replace(",", " ").split()
            if len(f) > 1 and f[0].startswith("s_") and cons:
                dst = sreg_set(f[1])
                if dst:
                    keep, dead = set(), False
                    for (reg, want) in cons:
                        if sreg_set(reg) & dst:
                            if f[0] == "s_mov_b64" and f[1] == reg and len(f) == 3 and f[2] in ("0", "-1"):
                                if int(f[2]) != want:
                                    dead = True          # this path contradicts itself: not a path
                            # (any other writer: the value is unknown, the constraint is dropped)
                        else:
                            keep.add((reg, want))
                    if dead:
                        continue
                    cons = frozenset(keep)
            verdict, st2 = visit(j, st)
            if verdict is True:
                continue
            if isinstance(verdict, str):
                return verdict
            key = (j, st2, cons)
            if key in seen:
                continue
            seen.add(key)
            if not self.pred[j]:
                v = entry_verdict(st2)
                if isinstance(v, str):
                    return v
                continue
            stack.extend((p, j, st2, cons) for p in self.pred[j])
        return None


def scan_kernel(name, ins, lab):
    findings = []
    n_slots = a_slots(name)
    if n_slots is None:
        return [f"{name}: cannot read the tile form from the symbol"]
    if not any(is_lds_dma(t) for t, _ in ins):
        return [f"{name}: no LDS-DMA request found (global_load_lds*): is this still the kernel the scan was written for?"]
    w = Walker(ins, lab)

    # 1. the counted waits
    n_waits = 0
    for i, (t, in_asm) in enumerate(ins):
        if not (in_asm and t.startswith("s_waitcnt")):
            continue
        m = re.search(r"vmcnt\((\d+)\)", t)
        if not m or int(m.group(1)) == 0:
            continue
        n = int(m.group(1))
        n_waits += 1
        if n != n_slots:
            findings.append(f"{name}: counted wait vmcnt({n}) at instruction {i}, but the tile form has A_SLOTS = {n_slots}")
            continue

        def visit(j, c, n=n):
            t2 = ins[j][0]
            op2 = t2.split()[0]
            if not is_vmem(op2):
                return None, c
            if is_lds_dma(t2):
                return (f"an LDS-DMA request (instruction {j}: `{t2}`) is among the {n} youngest requests on a path into the wait "
                        f"({c} activation loads behind it)"), c
            if not op2.startswith("buffer_load_dwordx4"):
                return f"`{t2}` (instruction {j}) is among the {n} youngest requests on a path into the wait", c
            return (True if c + 1 == n else None), c + 1

        bad = w.walk(i, visit, 0, lambda c: None)      # (a path from the entry with fewer loads has no LDS-DMA request in flight either)
        if bad:
            findings.append(f"{name}: s_waitcnt vmcnt({n}) at instruction {i}: {bad}")
    if n_waits == 0:
        findings.append(f"{name}: no counted inline-asm wait (s_waitcnt vmcnt(N), N > 0) found")

    # 2. a counted / draining wait of the kernel (inline asm) hands over at a barrier: `s_waitcnt vmcnt(..)` -> [`s_waitcnt lgkmcnt(0)`] ->
    # `s_barrier`, nothing that touches LDS or memory in between
    for i, (t, in_asm) in enumerate(ins):
        if not (in_asm and t.startswith("s_waitcnt") and "vmcnt" in t):
            continue
        j, ok = i + 1, False
        while j < len(ins) and j - i <= 6:
            op2 = ins[j][0].split()[0]
            if op2 == "s_barrier":
                ok = True
                break
            if not op2.startswith("s_") or op2.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_load", "s_buffer_load")):
                break                      # (other scalar ALU instructions the scheduler drops in between are harmless)
            j += 1
        if not ok:
            findings.append(f"{name}: the kernel's wait `{t}` at instruction {i} is not followed by its barrier")

    # 3. no LDS-DMA request may still be in flight when the kernel's LDS is reused (the staged epilogue) or the wave ends: walking backwards
    # from every s_endpgm, a covering wait is met before any request
    for i, (t, _) in enumerate(ins):
        if t.split()[0] != "s_endpgm":
            continue

        def visit(j, st):
            t2, in_asm = ins[j]
            op2 = t2.split()[0]
            if op2 == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", t2)
                if m and (int(m.group(1)) == 0 or (in_asm and int(m.group(1)) == n_slots)):
                    return True, st
            if is_vmem(op2) and is_lds_dma(t2):
                return f"LDS-DMA request at instruction {j} (`{t2}`) can reach the end of the kernel with no covering wait behind it", st
            return None, st

        bad = w.walk(i, visit, 0, lambda st: None)
        if bad:
            findings.append(f"{name}: s_endpgm at instruction {i}: {bad}")
    return findings


def main(paths):
    total, findings = 0, []
    for p in paths:
        kernels, labels = parse(p)
        for name, ins in kernels.items():
            total += 1
            findings += scan_kernel(name, ins, labels[name])
    for f in findings:
        print("tap8_pipeline_scan:", f)
    print(f"tap8_pipeline_scan: {total} tap_gemm8_kernel instantiations, {len(findings)} findings")
    if total == 0:
        print("tap8_pipeline_scan: no tap_gemm8_kernel in the given assembly", file=sys.stderr)
        return 1
    return 1 if findings else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
