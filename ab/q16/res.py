"""register / spill / scratch report of one q16 / multi-level translation unit for a set of -D flags (diagnostic):
   python ab/q16/res.py fused_q1.hip [-DFOO ...]   -> one line per training kernel (MODE 1)"""
import re, subprocess, sys
src, flags = sys.argv[1], sys.argv[2:]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-slp-vectorize",
       "-Wno-unused-function", *flags, "-Rpass-analysis=kernel-resource-usage", "-c", "neural_image_compression_v2_amd/csrc/" + src, "-o", "/dev/null"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
if "error" in out:
    print(out[:3000])
cur = None
rows = {}
for l in out.split("\n"):
    m = re.search(r"Function Name: (\S+)", l)
    if m:
        cur = m.group(1); rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z /\[\]]+): (\S+)", l)
    if m and cur:
        rows[cur][m.group(1).strip()] = m.group(2)
for k, v in rows.items():
    m = re.search(r"fused_q16_kernelINS_(\w+?)EEELi(\d)ELi(\d)", k)
    if not m or m.group(2) != "1":
        continue
    print(" ".join(flags), m.group(1), "NL", m.group(3), "VGPRs", v.get("VGPRs"), "spill", v.get("VGPRs Spill"), "SGPR spill", v.get("SGPRs Spill"), "scratch", v.get("ScratchSize [bytes/lane]"), "LDS", v.get("LDS Size [bytes/block]"))
