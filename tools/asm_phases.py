#!/usr/bin/env python3
"""Static instruction mix of one kernel per source-line range (needs -g1 in the compile: .loc markers).
usage: asm_phases.py file.s kernel_substring file_basename  "name:lo-hi,name:lo-hi,..."   (lines of the INNERMOST inlined-at frame that lies in <file_basename>)"""
import collections
import re
import sys

path, key, base, spec = sys.argv[1:5]
ranges = []
for item in spec.split(','):
    n, r = item.split(':')
    lo, hi = r.split('-')
    ranges.append((n, int(lo), int(hi)))
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if re.match(r'^[_A-Za-z0-9]+:', l) and key in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
cur = None
count = collections.defaultdict(collections.Counter)
for l in lines[start + 1:end]:
    t = l.strip()
    if t.startswith('.loc'):
        # all frames "file:line:col" in the comment; take the outermost-most frame inside `base` (the kernel body's own line)
        frames = re.findall(r'([\w./-]+):(\d+):\d+', t)
        mine = [int(ln) for f, ln in frames if f.endswith(base)]
        cur = mine[-1] if mine else cur
        continue
    if not t or t.startswith('.') or t.startswith(';') or t.endswith(':'):
        continue
    op = t.split()[0]
    cls = ('valu_f64' if re.match(r'v_(fma|mul|add|fmac|div|rcp|max|min|rsq|sqrt|ldexp|frexp|trig|fract|cvt)\w*f64', op) else
           'lds' if op.startswith('ds_') else 'vmem_st' if 'store' in op else 'vmem_ld' if op.startswith(('global_load', 'scratch_load', 'buffer_load')) else
           'salu' if op.startswith('s_') else 'valu_other')
    name = next((n for n, lo, hi in ranges if cur is not None and lo <= cur <= hi), 'other')
    count[name][cls] += 1
cols = ['valu_f64', 'valu_other', 'salu', 'lds', 'vmem_ld', 'vmem_st']
print('%-22s' % 'phase' + ''.join('%11s' % c for c in cols) + '      total')
tot = collections.Counter()
for n, _, _ in ranges + [('other', 0, 0)]:
    c = count[n]
    tot.update(c)
    print('%-22s' % n + ''.join('%11d' % c[k] for k in cols) + '%11d' % sum(c.values()))
print('%-22s' % 'sum' + ''.join('%11d' % tot[k] for k in cols) + '%11d' % sum(tot.values()))
