"""diagnostic: per-phase cycle shares of the fused training kernel (needs a library built with -DNIC_STAMPS)"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from neural_image_compression_v2_amd import _lib, fused, fp_def
from neural_image_compression_v2_amd.image_compression import ColorDecoder
METHOD = int(os.environ.get("METHOD", "1"))            # 1: the 4K 2D workload; 3 / 4: a 128^3 volume
dev = torch.device("cuda:0")
torch.manual_seed(0)
if METHOD == 1:
    H, W = 2160, 3840
    NS = H * W
    fp, _ = fp_def.create_pyramid((H // 4, W // 4), 12, 8, dev, torch.float32, True)
    dec = ColorDecoder(73, 64).to(dev)
    geo = fused.PathGeometry(dim=2, method=1, step_number=0.25, mip_level=0, extent=(H, W), num_crops=1,
                             noise_mode=int(os.environ.get("NOISE", "2")), noise_seed=7, noise_offset=1,
                             flags=_lib.NIC_FLAG_SPLIT_BF16 if os.environ.get("SPLIT") == "1" else 0)
    org = torch.zeros(1, 2, dtype=torch.int32, device=dev)
else:
    S = 128
    NS = S ** 3
    fp, _ = fp_def.create_pyramid_3d(S // 4, 12, 8, dev, torch.float32, True)
    dec = ColorDecoder(127 if METHOD == 3 else 79, 64).to(dev)
    geo = fused.PathGeometry(dim=3, method=METHOD, step_number=0.25, mip_level=0, extent=(S, S, S), num_crops=1,
                             noise_mode=int(os.environ.get("NOISE", "2")), noise_seed=7, noise_offset=1)
    org = torch.zeros(1, 3, dtype=torch.int32, device=dev)
params = [p.detach() for p in dec.linear_params()]
target = torch.rand(NS, 3, device=dev)
for _ in range(3):
    out = fused.fused_forward_backward(geo, fp[0].detach(), fp[1].detach(), org, params, target)
torch.cuda.synchronize()
d = geo.to_desc(fp[0], fp[1])
ws = _lib.workspace(dev, int(_lib.load().nic_workspace_bytes(ctypes.byref(d))))
KT = {1: 3, 3: 4, 4: 3}[METHOD]
NACC = 2 * KT + 4
PART = KT & 1
PART16 = PART and METHOD == 1
REC = (NACC + (4 if PART16 else 8 if PART else 0)) * 1024 + 4 * 320        # Lds<L>::REC (fused_kernel.hpp)
off = 256 * REC * 4
st = ws[off:off + 1024 * 16 * 8].view(torch.int64).view(1024, 16).cpu().numpy().astype(np.float64)
names = ["0 encode+noise", "1 X^T store, L1, L2, GELUs", "2 L3, dZ3, dW3 pass, dA2", "3 dZ2^T/A1^T stores, db2", "4 wait barrier 1", "5 dW2 MFMAs",
         "6 wait barrier 2", "7 dA1, dZ1, dZ1^T store", "8 wait barrier 3", "9 dW1 MFMAs", "10 wait barrier 4", "11 dX MFMAs + grid acc",
         "12 macro-tile setup", "13 grid flush"]
tot = st[:, :14].sum(1)
print(f"waves {st.shape[0]}, total cycles/wave median {np.median(tot):.3e} (min {tot.min():.3e}, max {tot.max():.3e})")
rounds = NS / 32 / 1024
for i, n in enumerate(names):
    print(f"{n:34s} {100 * np.median(st[:, i] / tot):6.2f} %   {np.median(st[:, i]) / rounds:9.0f} cycles/round")
