"""diagnostic: per-phase cycle shares of fused_train16_kernel on the 4K workload (needs a library built with -DNIC_STAMPS:
ab/w16/mk.sh st -DNIC_STAMPS; NIC_LIB_PATH=ab/libst.so python ab/w16/stamps16.py)"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from neural_image_compression_v2_amd import _lib, fused, fp_def
from neural_image_compression_v2_amd.image_compression import ColorDecoder
dev = torch.device("cuda:0")
torch.manual_seed(0)
H, W = 2160, 3840
NS = H * W
fp, _ = fp_def.create_pyramid((H // 4, W // 4), 12, 8, dev, torch.float32, True)
dec = ColorDecoder(73, 64).to(dev)
geo = fused.PathGeometry(dim=2, method=1, step_number=0.25, mip_level=0, extent=(H, W), num_crops=1,
                         noise_mode=int(os.environ.get("NOISE", "2")), noise_seed=7, noise_offset=1,
                         flags=_lib.NIC_FLAG_ORIGINS_ALIGNED, split_bf16=True)
org = torch.zeros(1, 2, dtype=torch.int32, device=dev)
params = [p.detach() for p in dec.linear_params()]
target = torch.rand(NS, 3, device=dev)
for _ in range(3):
    out = fused.fused_forward_backward(geo, fp[0].detach(), fp[1].detach(), org, params, target)
torch.cuda.synchronize()
d = geo.to_desc(fp[0], fp[1])
ws = _lib.workspace(dev, int(_lib.load().nic_workspace_bytes(ctypes.byref(d))))
REC, NWG = 20992, 256
off = NWG * REC * 4
st = ws[off:off + NWG * 8 * 16 * 8].view(torch.int64).view(NWG * 8, 16).cpu().numpy().astype(np.float64)
names = ["0 coords, blend, PE, noise", "1 L1, L2, L3 (+ image stores, GELUs)", "2 dZ3 image, dW3, dA2", "3 dA1 (+ dZ2 image), db2", "4 wait barrier 1",
         "5 dW2 MFMAs", "6 wait barrier 2", "7 dX (+ dZ1 image), grid acc", "8 wait barrier 3", "9 dW1 MFMAs", "10 wait barrier 4", "11 -",
         "12 macro-tile setup", "13 grid flush"]
tot = st[:, :14].sum(1)
print(f"waves {st.shape[0]}, total cycles/wave median {np.median(tot):.3e} (min {tot.min():.3e}, max {tot.max():.3e})")
rounds = NS / 16 / (NWG * 8)
for i, n in enumerate(names):
    print(f"{n:40s} {100 * np.median(st[:, i] / tot):6.2f} %   {np.median(st[:, i]) / rounds:9.0f} cycles/round")
wave_id = np.arange(st.shape[0]) % 8
for kh in (0, 1):
    sel = (wave_id >> 2) == kh
    print(f"-- half {kh}: total cycles/wave median {np.median(tot[sel]):.3e}")
    for i, n in enumerate(names):
        print(f"   {n:40s} {np.median(st[sel, i]) / rounds:9.0f} cycles/round")
