"""Print (or histogram) one basic block of one kernel in a hipcc -S listing.  usage: isa_block_dump.py file.s <name substring> <label e.g. .LBB20_84> [hist]"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
m = re.search(r'^(_Z\S*%s\S*):' % re.escape(sys.argv[2]), s, re.M)
body = s[m.end():s.index('.Lfunc_end', m.end())]
i = body.index(sys.argv[3] + ':')
j = re.search(r'^\.LBB\d+_\d+:', body[i + 5:], re.M)
blk = body[i:i + 5 + j.start()] if j else body[i:]
lines = [l.strip() for l in blk.split('\n') if l.strip() and not l.strip().startswith(';')]
if len(sys.argv) > 4:
    c = collections.Counter(l.split()[0] for l in lines[1:])
    for k, v in c.most_common(): print(v, k)
else:
    print('\n'.join(lines))
