"""Instruction counts per inner loop of one kernel in a -save-temps assembly file (a scheduling / instruction-count aid).
usage: python tools/isa_loops.py <file.s> <substring of the mangled kernel name> [min_instrs]"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
names = [n for n in re.findall(r'^(_ZN4hmcg\S*):', s, re.M) if sys.argv[2] in n]
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 100
for n in names[:1]:
    i = s.index('\n' + n + ':')
    j = s.index('.Lfunc_end', i)
    blocks, cur = [], None
    for ln in s[i:j].split('\n'):
        m = re.match(r'^(\.LBB\d+_\d+):(.*)', ln)
        if m:
            cur = [m.group(1), [], m.group(2)]
            blocks.append(cur)
            continue
        t = ln.strip()
        if cur is None:
            continue
        if t.startswith(';') and ('Loop' in t or 'Header' in t):
            cur[2] += ' ' + t
        if t and not t.startswith(';') and not t.startswith('.'):
            cur[1].append(t.split()[0])
    groups = collections.OrderedDict()
    for name, ins, c in blocks:
        m = re.findall(r'Header=(BB\d+_\d+) Depth=(\d)', c)
        hdr = None
        if 'Loop Header: Depth=2' in c:
            hdr = name[2:]
        elif m:
            d2 = [h for h, d in m if d == '2']
            hdr = d2[0] if d2 else None
        if hdr:
            groups.setdefault(hdr, []).append((name, len(ins), collections.Counter(ins)))
    print(n)
    for h, bl in groups.items():
        tot = sum(b[1] for b in bl)
        cc = collections.Counter()
        for b in bl:
            cc += b[2]
        if tot >= minn:
            print(' ', h, len(bl), 'blocks', tot, 'instrs', cc.most_common(12))
