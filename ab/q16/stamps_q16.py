"""diagnostic: per-phase cycle shares of fused_q16_kernel on the 4K workload (library built with -DNIC_STAMPS: ab/q16/mk.sh st -DNIC_STAMPS;
   NIC_LIB_PATH=ab/libst.so python ab/q16/stamps_q16.py [NL] [grid dtype: f32 | bf16])"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from neural_image_compression_v2_amd import _lib, fused, fp_def
from neural_image_compression_v2_amd.image_compression import ColorDecoder
NL = int(sys.argv[1]) if len(sys.argv) > 1 else 3
gdt = torch.bfloat16 if len(sys.argv) > 2 and sys.argv[2] == "bf16" else torch.float32
dev = torch.device("cuda:0")
torch.manual_seed(0)
H, W = 2160, 3840
NS = H * W
fp, _ = fp_def.create_pyramid((H // 4, W // 4), 12, 8, dev, torch.float32, True)
g0, g1 = fp[0].detach().to(gdt), fp[1].detach().to(gdt)
dec = ColorDecoder(73, 64, NL).to(dev)
geo = fused.PathGeometry(dim=2, method=1, step_number=0.25, mip_level=0, extent=(H, W), num_crops=1,
                         noise_mode=int(os.environ.get("NOISE", "2")), noise_seed=7, noise_offset=1,
                         flags=_lib.NIC_FLAG_ORIGINS_ALIGNED, bf16=True)
org = torch.zeros(1, 2, dtype=torch.int32, device=dev)
params = [p.detach() for p in dec.linear_params()]
target = torch.rand(NS, 3, device=dev)
for _ in range(3):
    out = fused.fused_forward_backward(geo, g0, g1, org, params, target)
torch.cuda.synchronize()
d = geo.to_desc(g0, g1)
ws = _lib.workspace(dev, int(_lib.load().nic_workspace_bytes(ctypes.byref(d))))
NH = NL - 2
NACC = (NH + 2) // 2
REC = 8 * NACC * 1024 + 8 * 256 + 8 * (NH * 64 + 192 + 4)
NWG = 256
off = NWG * REC * 4
st = ws[off:off + NWG * 8 * 16 * 8].view(torch.int64).view(NWG * 8, 16).cpu().numpy().astype(np.float64)
names = {0: "encode + noise", 1: "forward layers (+ image stores, GELUs)", 2: "dW_out, dA_last", 11: "dX (+ dZ1 image), grid sums", 12: "macro-tile setup + gather",
         13: "flush", 14: "barrier + dW1 (+ tail)", 15: "round-end barrier"}
for j in range(NH):
    names[3 + 2 * j] = f"phase {j}: dZ image, dA, db"
    names[4 + 2 * j] = f"phase {j}: barrier + owned dW"
tot = st.sum(1)
print(f"NL {NL}: waves {st.shape[0]}, total cycles/wave median {np.median(tot):.3e} (min {tot.min():.3e}, max {tot.max():.3e})")
rounds = NS / 16 / (NWG * 8)
wave_id = np.arange(st.shape[0]) % 8
h0, h1 = (wave_id >> 2) == 0, (wave_id >> 2) == 1
print(f"{'phase':46s} {'share':>7s} {'cyc/round':>10s} {'half 0':>9s} {'half 1':>9s}")
for i in sorted(names):
    print(f"{i:2d} {names[i]:43s} {100 * np.median(st[:, i] / tot):6.2f}% {np.median(st[:, i]) / rounds:10.0f} {np.median(st[h0, i]) / rounds:9.0f} {np.median(st[h1, i]) / rounds:9.0f}")
print(f"{'total':46s} {'':7s} {np.median(tot) / rounds:10.0f} {np.median(tot[h0]) / rounds:9.0f} {np.median(tot[h1]) / rounds:9.0f}")
