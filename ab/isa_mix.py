"""instruction mix of a kernel's ISA between labels (diagnostic): python ab/isa_mix.py file.s [start_line end_line]"""
import re, sys, collections
lines = open(sys.argv[1]).read().split("\n")
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (0, len(lines))
# basic-block labels and backward branches
labels = {m.group(1): i for i, l in enumerate(lines) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
if len(sys.argv) <= 3:
    for i, l in enumerate(lines):
        m = re.match(r"\s+(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)", l)
        if m and m.group(2) in labels and labels[m.group(2)] < i:
            print(f"backward branch line {i+1} -> {m.group(2)} (line {labels[m.group(2)]+1}), span {i - labels[m.group(2)]}")
def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_accvgpr"): return "accvgpr"
    if op in ("v_exp_f32", "v_rcp_f32", "v_log_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32"): return "trans"
    if op.startswith("v_pk_"): return "valu_pk"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem_atomic" if "atomic" in op else "vmem"
    if op == "s_waitcnt": return "waitcnt"
    if op == "s_nop": return "nop"
    if op.startswith("s_"): return "salu"
    return "other"
c = collections.Counter(); ops = collections.Counter()
for l in lines[lo:hi]:
    m = re.match(r"^\s+([a-z_0-9]+)", l)
    if not m or l.strip().startswith((".", ";")): continue
    c[cls(m.group(1))] += 1; ops[m.group(1)] += 1
print(dict(c))
print(ops.most_common(40))
