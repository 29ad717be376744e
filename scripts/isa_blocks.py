"""Instruction mix per basic block of one kernel in a hipcc -S listing.
usage: isa_blocks.py <file.s> <kernel-name-substring> [min_instructions]"""
import collections
import re
import sys

src, key = sys.argv[1], sys.argv[2]
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 8
lines = open(src).read().split('\n')
start = next(i for i, l in enumerate(lines) if key in l and re.match(r'^[A-Za-z_][^ \t]*:', l))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('s_endpgm'))
print(lines[start][:100], 'lines', start, end)


def cls(op):
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith('ds_read') or op.startswith('ds_load'): return 'lds_r'
    if op.startswith('ds_write') or op.startswith('ds_store'): return 'lds_w'
    if op.startswith(('buffer_load', 'global_load', 'flat_load', 'scratch_load')): return 'vm_r'
    if op.startswith(('buffer_store', 'global_store', 'flat_store', 'scratch_store')): return 'vm_w'
    if op.startswith('v_'): return 'valu'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith('s_barrier'): return 'barrier'
    if op.startswith('s_cbranch') or op.startswith('s_branch'): return 'branch'
    if op.startswith('s_'): return 'salu'
    return 'other'


blocks, cur, name = [], collections.Counter(), 'entry'
for l in lines[start + 1:end + 1]:
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.'):
        m = re.match(r'^(\.LBB[0-9_]+):', t)
        if m:
            blocks.append((name, cur)); cur, name = collections.Counter(), m.group(1)
        continue
    op = t.split()[0]
    cur[cls(op)] += 1
    if op.startswith('s_cbranch') or op.startswith('s_branch'):
        cur['->' + t.split()[-1]] += 0
blocks.append((name, cur))
tot = collections.Counter()
for n, c in blocks:
    for k, v in c.items():
        tot[k] += v
    n_ins = sum(v for k, v in c.items() if not k.startswith('->'))
    if n_ins >= minn:
        tgt = ' '.join(k for k in c if k.startswith('->'))
        print(f'{n:14s} n={n_ins:5d} ' + ' '.join(f'{k}={c[k]}' for k in ('mfma', 'valu', 'salu', 'lds_r', 'lds_w', 'vm_r', 'vm_w',
                                                                     'wait', 'barrier') if c[k]) + ' ' + tgt)
print('total', {k: v for k, v in tot.items() if not k.startswith('->')})
