"""diagnostic: per-phase cycles of fused_q16_kernel on the reference's 3D sweep step (8 random 32^3 crops of a 64^3 volume); library built with -DNIC_STAMPS
   NIC_LIB_PATH=ab/libst.so python ab/q16/stamps_sweep.py [method]"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from neural_image_compression_v2_amd import _lib, fused, fp_def
from neural_image_compression_v2_amd.image_compression import ColorDecoder
method = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda:0")
torch.manual_seed(0)
S, crop = 64, 32
fp, _ = fp_def.create_pyramid_3d(S // 4, 12, 8, dev, torch.float32, True)
g0, g1 = fp[0].detach(), fp[1].detach()
cin = 127 if method == 3 else 79
dec = ColorDecoder(cin, 64, 3).to(dev)
rs = np.random.RandomState(0)
org = [tuple(int(rs.randint(0, S - crop + 1)) for _ in range(3)) for _ in range(8)]
geo = fused.PathGeometry(dim=3, method=method, step_number=0.25, mip_level=0, extent=(crop,) * 3, num_crops=8, noise_mode=2, noise_seed=7, noise_offset=1, bf16=True)
params = [p.detach() for p in dec.linear_params()]
target = torch.rand(8 * crop ** 3, 3, device=dev)
for _ in range(3):
    out = fused.fused_forward_backward(geo, g0, g1, org, params, target)
torch.cuda.synchronize()
d = geo.to_desc(g0, g1)
ws = _lib.workspace(dev, int(_lib.load().nic_workspace_bytes(ctypes.byref(d))))
NACC = 2 if method == 3 else 1
REC = 8 * NACC * 1024 + (8 * 256 if method == 4 else 0) + 8 * (64 + 192 + 4)
NWG = 256
off = NWG * REC * 4
st = ws[off:off + NWG * 8 * 16 * 8].view(torch.int64).view(NWG * 8, 16).cpu().numpy().astype(np.float64)
names = {9: "prologue: W1 image", 7: "prologue: hidden + output images", 8: "prologue: biases", 5: "prologue: zeroing", 6: "prologue: barrier + accumulators", 10: "record writes", 0: "encode + noise", 1: "forward", 2: "dW_out, dA_last", 3: "phase 0 compute", 4: "phase 0 barrier + dW", 11: "dX + grid sums", 12: "unit setup + gather", 13: "flush (combine, group sum, atomics)",
         14: "barrier + dW1", 15: "round-end barrier"}
tot = st.sum(1)
print(f"method {method}: total cycles per wave median {np.median(tot):.0f} = {np.median(tot) / 2.4e3:.1f} us at 2.4 GHz (min {tot.min():.0f}, max {tot.max():.0f})")
for i in sorted(names):
    print(f"{i:2d} {names[i]:40s} {100 * np.median(st[:, i] / np.maximum(tot, 1)):6.2f}%  {np.median(st[:, i]):9.0f} cycles per wave")
