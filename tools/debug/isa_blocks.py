"""Instruction-class counts per basic block of one kernel in a hipcc -S listing.  usage: isa_blocks.py file.s <substring of mangled name> [min ops]"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
m = re.search(r'^(_Z\S*%s\S*):' % re.escape(sys.argv[2]), s, re.M)
start = m.end()
end = s.index('.Lfunc_end', start)
body = s[start:end]
minops = int(sys.argv[3]) if len(sys.argv) > 3 else 40


def cat(op):
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith('ds_'): return 'lds'
    if op.startswith('scratch_'): return 'scratch'
    if op.startswith(('global_', 'buffer_', 'flat_')): return 'vmem'
    if op.startswith(('v_readlane', 'v_readfirstlane', 'v_writelane')): return 'lanex'
    if 'dpp' in op or op.startswith(('v_permlane', 'ds_bpermute', 'ds_swizzle')): return 'xlane'
    if op.startswith('v_'): return 'valu'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith('s_barrier'): return 'barrier'
    if op.startswith('s_'): return 'salu'
    return 'other'


blocks, cur = [], ('entry', [])
for l in body.split('\n'):
    t = l.strip()
    if not t or t.startswith(';'): continue
    if re.match(r'^\.LBB\d+_\d+:', t):
        blocks.append(cur); cur = (t, []); continue
    if t.startswith('.'): continue
    op = t.split()[0]
    if 'dpp' in t and op.startswith('v_'): op = op + '_dpp'
    cur[1].append(op)
blocks.append(cur)
tot = collections.Counter()
for n, ops in blocks:
    c = collections.Counter(cat(o) for o in ops)
    tot += c
    if len(ops) >= minops: print(n, len(ops), dict(c))
print('total', dict(tot))
