#!/usr/bin/env python3
"""Approximate VGPR pressure profile of one kernel in an AMDGPU assembly listing (hipcc --save-temps: *.s).

Linear-order liveness: a register is counted as live between its first write and its last read in program order (loops and
branches are ignored, so values carried around a loop are under-counted).  Good enough to see WHERE a kernel holds its registers:
prints the pressure every `step` instructions together with the source line markers (.loc) nearest to the peak.

usage: asm_pressure.py file.s kernel_substring [step]
"""
import re
import sys


def regs_of(tok):
    out = []
    for m in re.finditer(r'\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b', tok):
        if m.group(1):
            out += [(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)]
        else:
            out.append((m.group(4), int(m.group(5))))
    return out


def main():
    path, key = sys.argv[1], sys.argv[2]
    step = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    at = int(sys.argv[4]) if len(sys.argv) > 4 else None      # list the live ranges that cover this instruction
    lines = open(path).read().split('\n')
    start = next(i for i, l in enumerate(lines) if re.match(r'^[_A-Za-z0-9]+:', l) and key in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
    insts, locs = [], []
    loc = ''
    for l in lines[start + 1:end]:
        t = l.strip()
        if t.startswith('.loc'):
            loc = t
            continue
        if not t or t.startswith('.') or t.startswith(';') or t.endswith(':'):
            continue
        t = t.split(';')[0].strip()
        m = re.match(r'(\S+)\s*(.*)', t)
        op, args = m.group(1), m.group(2)
        ops = [a.strip() for a in args.split(',')] if args else []
        if op.startswith(('global_store', 'ds_write', 'scratch_store', 'buffer_store', 's_', 'v_cmp', 'global_atomic')) and not op.startswith('v_cmpx'):
            dst, srcs = [], ops
            if op.startswith('v_cmp') and ops and ops[0].startswith(('s[', 'vcc')):
                srcs = ops[1:]
        else:
            dst, srcs = ops[:1], ops[1:]
            if op.startswith(('v_fmac', 'v_mac', 'v_accvgpr_write')) or 'dpp' in t and op.startswith('v_mov'):
                srcs = ops      # read-modify-write destinations
        insts.append((op, [r for o in dst for r in regs_of(o)], [r for o in srcs for r in regs_of(o)]))
        locs.append(loc)
    first_w, last_r = {}, {}
    for i, (op, d, s) in enumerate(insts):
        for r in d:
            first_w.setdefault(r, i)
        for r in s:
            last_r[r] = i
            first_w.setdefault(r, 0)
    # re-definitions: split live ranges at each write that is not preceded by a read of the old value since the previous write
    events = [0] * (len(insts) + 1)
    live = {}
    ranges = []
    for i, (op, d, s) in enumerate(insts):
        for r in s:
            if r in live:
                live[r][1] = i
        for r in d:
            if r in live and r not in s:
                ranges.append(tuple(live[r]))
            if r not in live or r not in s:
                live[r] = [i, i]
    ranges += [tuple(v) for v in live.values()]
    if at is not None:
        import collections
        short = lambda l: ' <- '.join(re.findall(r'(\w+\.h:\d+|\w+\.hip:\d+)', l)[:3])
        by_def = collections.Counter((short(locs[a]), short(locs[b])) for a, b in ranges if a < at <= b)
        for (d, u), n in sorted(by_def.items(), key=lambda kv: -kv[1])[:60]:
            print('%3d regs  defined %-60s last use %s' % (n, d, u))
        return
    for a, b in ranges:
        events[a] += 1
        events[b] -= 1
    cur, prof = 0, []
    for e in events[:-1]:
        cur += e
        prof.append(cur)
    peak = max(range(len(prof)), key=lambda i: prof[i])
    print('instructions %d, peak pressure %d at instruction %d (%s)' % (len(insts), prof[peak], peak, locs[peak]))
    for i in range(0, len(prof), step):
        seg = prof[i:i + step]
        j = i + max(range(len(seg)), key=lambda k: seg[k])
        print('%6d..%6d  max %3d  %s' % (i, i + len(seg) - 1, prof[j], locs[j]))


if __name__ == '__main__':
    main()
