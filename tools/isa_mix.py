#!/usr/bin/env python3
"""Instruction mix of the loops of one kernel in a hipcc -S listing.  Usage: isa_mix.py file.s mangled_name_substring"""
import re, sys, collections
s = open(sys.argv[1]).read()
key = sys.argv[2]
m = re.search(r'^(\S*' + re.escape(key) + r'\S*):', s, re.M)
i = m.start()
j = s.index('.end_amdhsa_kernel', i)
lines = []
for l in s[i:j].split('\n'):
    l = l.split(';')[0].strip()
    if not l: continue
    if l.startswith('.') and not l.endswith(':'): continue
    lines.append(l)
labels = {l[:-1]: n for n, l in enumerate(lines) if l.endswith(':')}
def grp(k):
    if k.startswith('v_') and 'f64' in k: return 'dp'
    if k.startswith('v_accvgpr'): return 'acc'
    if k.startswith('v_mov'): return 'v_mov'
    if k.startswith('v_cndmask'): return 'cndmask'
    if k.startswith('global_load'): return 'gload'
    if k.startswith('global_store'): return 'gstore'
    if k.startswith('ds_'): return 'ds'
    if k.startswith('s_waitcnt'): return 'waitcnt'
    if k.startswith('s_'): return 'salu'
    if k.startswith('v_'): return 'valu_other'
    return 'other'
print(len(lines), 'instructions+labels in kernel')
for n, l in enumerate(lines):
    mm = re.match(r's_c?branch\S*\s+(\S+)', l)
    if mm and mm.group(1) in labels and labels[mm.group(1)] < n:
        a = labels[mm.group(1)]
        seg = [x for x in lines[a:n] if not x.endswith(':')]
        g = collections.Counter(grp(x.split()[0]) for x in seg)
        print('loop', mm.group(1), 'len', len(seg), dict(g))
        if len(sys.argv) > 3:
            c = collections.Counter(x.split()[0] for x in seg)
            print(c.most_common(40))
            print([x for x in seg if x.startswith('s_waitcnt')])
