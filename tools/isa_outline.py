#!/usr/bin/env python
"""Outline of one kernel's gfx950 ISA: labels, branches, waits, MFMA / memory ops.
usage: isa_outline.py file.s mangled_prefix [--full]"""
import sys
lines = open(sys.argv[1]).read().split('\n')
pref = sys.argv[2]
start = [i for i, l in enumerate(lines) if l.startswith(pref) and ':' in l.split(';')[0]][0]
end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
body = lines[start:end + 1]
keys = ('s_cbranch', 's_waitcnt', 'v_mfma', 'global_load', 'ds_read', 'ds_write', 's_barrier', 'global_store',
        'scratch_', 'buffer_')
full = '--full' in sys.argv
print("lines", len(body))
run = None
cnt = 0
def flush():
    global run, cnt
    if run is not None:
        print(f"      {run} x{cnt}")
    run, cnt = None, 0
for i, l in enumerate(body):
    t = l.strip()
    if not t or t.startswith(';'):
        continue
    if t.startswith('.LBB') or t.startswith('s_cbranch') or t.startswith('s_branch'):
        flush(); print(i, t[:80]); continue
    op = t.split()[0]
    if full or any(op.startswith(k) for k in keys):
        if op == run:
            cnt += 1
        else:
            flush(); run, cnt = op, 1
    if op.startswith('s_waitcnt'):
        flush(); print(i, '   ', t[:60])
flush()
