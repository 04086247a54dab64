"""kernel-only timing of the 16-sample training kernel on the 4K workload: NOISE=0/2 (mode), TARGET=tensor/image"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from neural_image_compression_v2_amd import _lib, fused, fp_def
from neural_image_compression_v2_amd.image_compression import ColorDecoder
dev = torch.device("cuda:0")
torch.manual_seed(0)
H, W = 2160, 3840
fp, _ = fp_def.create_pyramid((H // 4, W // 4), 12, 8, dev, torch.float32, True)
dec = ColorDecoder(73, 64).to(dev)
params = [p.detach() for p in dec.linear_params()]
target = torch.rand(H * W, 3, device=dev)
org = torch.zeros(1, 2, dtype=torch.int32, device=dev)
for noise in [int(v) for v in os.environ.get("NOISE", "2,0").split(",")]:
    geo = fused.PathGeometry(dim=2, method=1, step_number=0.25, mip_level=0, extent=(H, W), num_crops=1, noise_mode=noise, noise_seed=7,
                             noise_offset=1, flags=_lib.NIC_FLAG_ORIGINS_ALIGNED, split_bf16=True)
    flat = None
    ts = []
    for i in range(40):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        out = fused.fused_forward_backward(geo, fp[0].detach(), fp[1].detach(), org, params, target, events=(a, b))
        torch.cuda.synchronize()
        if i >= 10: ts.append(a.elapsed_time(b))
    print(f"{os.environ.get('NIC_LIB_PATH', 'default').split('/')[-1]:24s} noise {noise}: median {np.median(ts):.4f} ms  min {min(ts):.4f}  loss {float(out.loss):.6f}")
