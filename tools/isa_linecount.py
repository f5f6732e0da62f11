#!/usr/bin/env python3
"""Static instruction count per source line of one kernel in a hipcc -save-temps -gline-tables-only .s file.

usage: isa_linecount.py file.s kernel_substring [lo hi]   (lo..hi: only lines of the main file in that range,
       inlined header lines are attributed to the last main-file line seen)
"""
import re, sys, collections
path, kern = sys.argv[1], sys.argv[2]
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 0
hi = int(sys.argv[4]) if len(sys.argv) > 4 else 10**9
files = {}
inside = False
cur = (0, 0)
anchor = 0
cnt = collections.Counter()
kinds = collections.defaultdict(collections.Counter)
for line in open(path):
    s = line.strip()
    m = re.match(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', s)
    if m:
        files[int(m.group(1))] = m.group(3) or m.group(2)
        continue
    if re.match(r'^[_A-Za-z0-9$.]+:', s) and not s.startswith('.L'):
        inside = kern in s
        continue
    if not inside:
        continue
    m = re.match(r'\.loc\s+(\d+)\s+(\d+)', s)
    if m:
        cur = (int(m.group(1)), int(m.group(2)))
        if files.get(cur[0], '').endswith('emd.hip') or files.get(cur[0], '').endswith(sys.argv[5] if len(sys.argv) > 5 else 'emd.hip'):
            anchor = cur[1]
        continue
    if s.startswith('.') or s.startswith(';') or not s:
        continue
    op = s.split()[0]
    if not re.match(r'^(v_|s_|ds_|buffer_|global_|flat_|scratch_)', op):
        continue
    if lo <= anchor <= hi:
        cnt[anchor] += 1
        k = 'valu' if op.startswith('v_') else 'salu' if op.startswith('s_') else 'lds' if op.startswith('ds_') else 'scratch' if op.startswith('scratch_') else 'vmem'
        kinds[anchor][k] += 1
tot = collections.Counter()
for ln in sorted(cnt):
    k = kinds[ln]
    print(f"{ln:5d} {cnt[ln]:5d}  valu {k['valu']:4d} salu {k['salu']:4d} lds {k['lds']:3d} vmem {k['vmem']:3d} scratch {k['scratch']:3d}")
    tot.update(k)
print('total', sum(cnt.values()), dict(tot))
