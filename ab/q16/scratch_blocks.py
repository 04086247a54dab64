"""per basic block of a kernel in an assembly listing: scratch loads / stores, MFMAs, instructions - where do the spills sit?
   python ab/q16/scratch_blocks.py /tmp/q1.s Li1ELi5"""
import re, sys
txt = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2]
start = None
for i, l in enumerate(txt):
    if l.startswith('_ZN3nic16fused_q16_kernel') and pat in l.split(':')[0]:
        start = i
        break
blocks, cur = [], ['entry', 0, 0, 0, 0]
blocks.append(cur)
for l in txt[start + 1:]:
    if l.startswith('.Lfunc_end') or '.end_amdhsa_kernel' in l:
        break
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        cur = [m.group(1), 0, 0, 0, 0]
        blocks.append(cur)
        continue
    t = l.strip()
    if t.startswith('scratch_load'): cur[1] += 1
    if t.startswith('scratch_store'): cur[2] += 1
    if t.startswith('v_mfma'): cur[3] += 1
    if t and not t.startswith(';') and not t.startswith('.'): cur[4] += 1
print('block, scratch loads, scratch stores, mfma, instructions')
for b in blocks:
    if b[1] + b[2] > 0 or b[3] > 0:
        print(' ', b)
