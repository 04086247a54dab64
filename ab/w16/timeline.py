"""ISA timeline of the first fused_train16 kernel in an assembly dump: runs of MFMAs, spills (scratch), barriers, atomics, gathers"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith('_ZN3nic20fused_train16_kernel'))
end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
body = lines[start:end]
events = []
counts = {}
for i, l in enumerate(body):
    t = l.strip()
    op = t.split(' ')[0] if t else ''
    counts[op.split('_e')[0]] = counts.get(op.split('_e')[0], 0) + 1
    if t.startswith('v_mfma'): e = 'M16' if '16x16x32' in t else ('M32' if '32x32' in t else 'M4')
    elif t.startswith('scratch_store'): e = 'ST'
    elif t.startswith('scratch_load'): e = 'LD'
    elif t.startswith('s_barrier'): e = 'BAR'
    elif t.startswith('global_atomic'): e = 'ATOM'
    elif t.startswith('global_load'): e = 'GL'
    elif t.startswith('s_cbranch'): e = 'BR'
    elif t.startswith('v_accvgpr'): e = 'ACC'
    else: continue
    events.append((i, e))
out = []; prev = None; cnt = 0; st = 0
for i, e in events:
    if e == prev: cnt += 1
    else:
        if prev: out.append(f"{st}:{prev}x{cnt}")
        prev = e; cnt = 1; st = i
out.append(f"{st}:{prev}x{cnt}")
print(len(body), 'lines')
print(' '.join(out))
top = sorted(counts.items(), key=lambda kv: -kv[1])[:40]
print(top)
