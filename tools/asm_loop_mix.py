"""Instruction mix of the hottest loop of a kernel: python tools/asm_loop_mix.py <kernel.s> <mangled-name prefix> ...
(produce the .s with: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -S --cuda-device-only -o k.s oi_k_cg1b.hip)"""
import re, sys
from collections import Counter
lines = open(sys.argv[1]).read().split('\n')
for prefix in sys.argv[2:]:
    start = next((i for i, l in enumerate(lines) if l.startswith(prefix) and ':' in l), None)
    if start is None:
        print("not found", prefix); continue
    end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
    body = lines[start:end]
    labels = [(i, l) for i, l in enumerate(body) if l.startswith('.LBB')]
    best = None
    for i, l in enumerate(body):
        m = re.search(r's_cbranch\w+\s+(\.LBB\d+_\d+)', l) or re.search(r's_branch\s+(\.LBB\d+_\d+)', l)
        if m:
            tgt = [j for j, ll in labels if ll.startswith(m.group(1) + ':')]
            if tgt and tgt[0] < i and i - tgt[0] < 4000 and (best is None or i - tgt[0] > best[1] - best[0]): best = (tgt[0], i)
    seg = [l.strip() for l in body[best[0]:best[1]] if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
    c = Counter()
    for l in seg:
        op = l.split()[0]
        if op.startswith('global_load'): c['gload'] += 1
        elif op.startswith('global_store'): c['gstore'] += 1
        elif op.startswith('ds_'): c['lds'] += 1
        elif 'f64' in op: c['f64'] += 1
        elif op.startswith('v_'): c['valu_other'] += 1
        elif op.startswith('s_waitcnt'): c['wait'] += 1
        elif op.startswith('s_'): c['salu'] += 1
        else: c[op] += 1
    print(prefix[:44], '| loop instrs', len(seg), dict(c))
