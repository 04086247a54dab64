"""run-length view of instruction classes (diagnostic): python ab/isa_runs.py file.s start end"""
import re, sys
sys.path.insert(0, __file__.rsplit("/", 1)[0])
lines = open(sys.argv[1]).read().split("\n")
lo, hi = int(sys.argv[2]), int(sys.argv[3])
def cls(op):
    if op.startswith("v_mfma"): return "M"
    if op.startswith("v_accvgpr"): return "a"
    if op in ("v_exp_f32_e32", "v_rcp_f32_e32"): return "T"
    if op.startswith("v_readlane") or op.startswith("v_writelane"): return "r"
    if op.startswith("v_"): return "v"
    if op.startswith("ds_read"): return "L"
    if op.startswith("ds_"): return "S"
    if "atomic" in op: return "A"
    if op.startswith(("global_load", "buffer_load")): return "G"
    if op.startswith(("global_store", "scratch_")): return "W"
    if op == "s_waitcnt": return "w"
    if op == "s_nop": return "n"
    if op == "s_barrier": return "B"
    if op.startswith("s_cbranch") or op == "s_branch": return "j"
    if op.startswith("s_"): return "s"
    return "?"
out = []; prev = None; cnt = 0; start = lo
for i in range(lo, hi):
    l = lines[i]
    m = re.match(r"^\s+([a-z_0-9]+)", l)
    if re.match(r"^\.LBB", l):
        if prev: out.append(f"{prev}{cnt}")
        out.append(f"\n[{i+1}:{l.strip()}]"); prev = None; cnt = 0; continue
    if not m or l.strip().startswith((".", ";")): continue
    c = cls(m.group(1))
    if c == prev: cnt += 1
    else:
        if prev: out.append(f"{prev}{cnt}")
        prev, cnt = c, 1
if prev: out.append(f"{prev}{cnt}")
print(" ".join(out))
