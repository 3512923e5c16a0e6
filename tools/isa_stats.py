#!/usr/bin/env python3
"""Instruction mix per basic block of the kernels in a hipcc -save-temps .s file (which loop issues what).
usage: isa_stats.py file.s [kernel-name-substring] [min-instructions-per-block]"""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ''
thr = int(sys.argv[3]) if len(sys.argv) > 3 else 100
for m in re.finditer(r'^(_Z\w+):.*\n', s, re.M):
    nm = m.group(1)
    if pat not in nm:
        continue
    i = m.end()
    j = s.find('.end_amdhsa_kernel', i)
    body = s[i:j if j > 0 else len(s)]
    print(nm[:80], 'vgpr', re.findall(r'\.amdhsa_next_free_vgpr (\d+)', body))
    blocks = re.split(r'\n(\.LBB\d+_\d+):', body)
    for k in range(1, len(blocks), 2):
        ins = re.findall(r'^\s+([vsdg][\w]+)', blocks[k + 1], re.M)
        if len(ins) > thr:
            print('  ', blocks[k], len(ins), Counter(ins).most_common(14))
