#!/usr/bin/env python3
"""List, per kernel of a gfx950 .s file, the vector-memory waits INSIDE loops: for every loop (the assembler's `in Loop: Header=BBx_y`
comments) the loads / stores / LDS-DMA issued in it and the `s_waitcnt vmcnt(N)` values met.  A `vmcnt(0)` in a loop that also issues
loads is the pattern round 5 found twice in the staged epilogue (a request waited for on the spot because hipcc turned a conditionally
loaded or rotated register into a copy).  Reporting tool, not a gate:  python tools/loop_waits.py <file.s> [kernel-substring ...]"""
import re, sys, collections

def kernels(lines):
    name, start = None, 0
    for i, l in enumerate(lines):
        m = re.match(r'^(_Z\w+):\s', l)
        if m:
            name, start = m.group(1), i
        elif name and 's_endpgm' in l:
            yield name, start, i
            name = None

def main():
    lines = open(sys.argv[1]).read().split('\n')
    pats = sys.argv[2:]
    for name, a, b in kernels(lines):
        if pats and not any(p in name for p in pats):
            continue
        cur = None
        loops = collections.OrderedDict()
        for i in range(a, b):
            l = lines[i]
            m = re.match(r'^\.LBB\d+_\d+:\s*;(.*)$', l)
            if m:
                c = m.group(1)
                h = re.search(r'Header=(BB\d+_\d+)', c)
                if h:
                    cur = h.group(1)
                elif 'Loop Header' in c:
                    cur = l.split(':')[0].lstrip('.L')
                    cur = 'BB' + cur[2:] if not cur.startswith('BB') else cur
                else:
                    cur = None
                continue
            if cur is None:
                continue
            t = l.strip()
            d = loops.setdefault(cur, collections.Counter())
            op = t.split(' ')[0] if t else ''
            if op.startswith(('buffer_load', 'global_load', 'flat_load')):
                d['lds_dma' if ' lds' in t else 'load'] += 1
            elif op.startswith(('buffer_store', 'global_store', 'flat_store')):
                d['store'] += 1
            elif op.startswith(('global_atomic', 'buffer_atomic', 'flat_atomic')):
                d['atomic'] += 1
            elif op == 's_waitcnt':
                m = re.search(r'vmcnt\((\d+)\)', t)
                if m:
                    d['vmcnt(%s)' % m.group(1)] += 1
            elif 'mfma' in op:
                d['mfma'] += 1
        rows = [(h, d) for h, d in loops.items() if any(k.startswith('vmcnt') for k in d)]
        if rows:
            print(name[:110])
            for h, d in rows:
                print('   loop %-12s %s' % (h, ' '.join('%s=%d' % kv for kv in sorted(d.items()))))

if __name__ == '__main__':
    main()
