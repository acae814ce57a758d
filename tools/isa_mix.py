#!/usr/bin/env python3
"""Instruction mix per kernel of a hipcc -S --cuda-device-only listing: tools/isa_mix.py file.s [name-substring]"""
import collections, re, sys
txt = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else ""
parts = re.split(r'\n(_Z[^\n:]*):[^\n]*\n', txt)
for i in range(1, len(parts), 2):
    name, body = parts[i], parts[i + 1].split('.Lfunc_end')[0]
    if want not in name or 'v_mfma' not in body:
        continue
    c = collections.Counter()
    for line in body.split('\n'):
        line = line.strip()
        if not line or line[0] in '.;/':
            continue
        c[line.split()[0]] += 1
    g = collections.Counter()
    for op, n in c.items():
        if op.startswith('v_mfma'): g['mfma'] += n
        elif op.startswith('v_accvgpr'): g[op] += n
        elif op.startswith('ds_'): g[op] += n
        elif op.startswith('s_waitcnt'): g['s_waitcnt'] += n
        elif op.startswith('s_nop'): g['s_nop'] += n
        elif op.startswith('v_'): g['valu'] += n
        elif op.startswith(('global_', 'buffer_', 'scratch_')): g[op] += n
        else: g['salu/other'] += n
    print(name[-60:], 'total', sum(c.values()))
    for k, v in sorted(g.items(), key=lambda x: -x[1]):
        print('   %-30s %6d  %.2f per mfma' % (k, v, v / g['mfma']))
    valu = collections.Counter({op: n for op, n in c.items() if op.startswith('v_') and not op.startswith(('v_mfma', 'v_accvgpr'))})
    print('   top valu:', valu.most_common(12))
