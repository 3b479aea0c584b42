#!/usr/bin/env python
"""Instruction mix of every loop of one kernel's gfx950 ISA (fp32 MFMA shares the SIMD's FMA lanes with the
VALU -- tools/micro/mfma_valu_overlap.hip -- so VALU instructions per MFMA is the figure to drive down).
usage: isa_loopmix.py file.s mangled_prefix [...]"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split('\n')


def analyze(pref):
    start = [i for i, l in enumerate(lines) if l.startswith(pref) and ':' in l][0]
    end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
    body = lines[start:end + 1]
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            labels[m.group(1)] = i
    loops = []
    for i, l in enumerate(body):
        m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    tot = collections.Counter()
    for l in body:
        t = l.strip()
        if t and not t.startswith((';', '.')):
            op = t.split()[0]
            tot['MFMA' if op.startswith('v_mfma') else 'VALU' if op.startswith('v_') else 'other'] += 1
    print(pref[:60], 'static totals', dict(tot))
    for (a, b) in loops:
        c = collections.Counter()
        for l in body[a:b + 1]:
            t = l.strip()
            if not t or t.startswith((';', '.')):
                continue
            op = t.split()[0]
            if op.startswith('v_mfma'):
                c['MFMA'] += 1
            elif op.startswith('v_'):
                c['VALU'] += 1
                c['  ' + op] += 1
            elif op.startswith(('s_waitcnt', 's_nop', 's_barrier')):
                c[op] += 1
            elif op.startswith('s_'):
                c['SALU'] += 1
            else:
                c[op] += 1
        if c['MFMA'] > 0 or c['VALU'] > 20:
            print('  loop', a, b, dict(c))


for p in sys.argv[2:]:
    analyze(p)
