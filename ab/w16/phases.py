"""instruction mix per stamped phase (s_memtime markers) of the first fused_train16 kernel in a -DNIC_STAMPS assembly dump"""
import sys, collections
lines = open(sys.argv[1]).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith('_ZN3nic20fused_train16_kernel'))
end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
body = lines[start:end]
def klass(op):
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith('ds_'): return 'lds'
    if op.startswith('global_') or op.startswith('scratch_') or op.startswith('buffer_'): return 'vmem'
    if op.startswith('v_readlane') or op.startswith('v_writelane'): return 'lane'
    if op.startswith('v_pk_'): return 'valu_pk'
    if op in ('v_exp_f32', 'v_rcp_f32', 'v_log_f32', 'v_sqrt_f32', 'v_rsq_f32'): return 'valu_tr'
    if op.startswith('v_cvt_pk_bf16'): return 'valu_cvt'
    if op.startswith('v_mov') or op.startswith('v_accvgpr'): return 'valu_mov'
    if op.startswith('v_'): return 'valu'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith('s_nop'): return 'nop'
    if op.startswith('s_barrier'): return 'barrier'
    if op.startswith('s_'): return 'salu'
    return 'other'
seg = collections.Counter(); segs = []; first = None
for i, l in enumerate(body):
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.') or t.endswith(':'): continue
    op = t.split()[0].replace('_e32', '').replace('_e64', '')
    if op == 's_memtime':
        segs.append((i, seg)); seg = collections.Counter(); continue
    seg[klass(op)] += 1
segs.append((len(body), seg))
keys = ['valu', 'valu_pk', 'valu_tr', 'valu_cvt', 'valu_mov', 'lane', 'mfma', 'lds', 'vmem', 'salu', 'wait', 'nop', 'barrier']
print('line   ' + ' '.join(f'{k:>8s}' for k in keys) + '    total')
for i, c in segs:
    print(f'{i:6d} ' + ' '.join(f'{c[k]:8d}' for k in keys) + f'   {sum(c.values()):6d}')
