"""Instruction mix of one kernel in a hipcc -S --cuda-device-only listing.  Usage: python tools/isamix.py file.s mangled_substring"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
pat = sys.argv[2]
m = re.search(r'^(_Z\S*' + re.escape(pat) + r'\S*):(.*?)s_endpgm', s, re.S | re.M)
print(m.group(1))
cnt = collections.Counter()
for l in m.group(2).split('\n'):
    l = l.strip()
    if not l or l.startswith(('.', ';')) or l.endswith(':'):
        continue
    op = l.split()[0]
    if op.startswith('v_mfma'):
        key = 'MFMA ' + op
    elif op.startswith('s_'):
        key = 'SALU ' + ('waitcnt' if 'waitcnt' in op else 'barrier' if 'barrier' in op else 'other')
    else:
        key = op
    cnt[key] += 1
print('total', sum(cnt.values()))
for k, v in cnt.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 30):
    print(f'{k:44s}{v}')
