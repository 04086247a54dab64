"""diagnostic: kernel time of the fused 2D step for small launches (num_crops x 256^2 random unaligned crops of a 512^2 image), HIP events"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from neural_image_compression_v2_amd import _lib, fused, fp_def
from neural_image_compression_v2_amd.image_compression import ColorDecoder
dev = torch.device("cuda:0")
torch.manual_seed(0)
S = 512
fp, _ = fp_def.create_pyramid((S // 4, S // 4), 12, 8, dev, torch.float32, True)
dec = ColorDecoder(73, 64).to(dev)
params = [p.detach() for p in dec.linear_params()]
import json
CFG = json.loads(os.environ.get('SMALL_CFG', '[[1, 16], [1, 64], [1, 256], [2, 256], [4, 256], [8, 256], [16, 256], [32, 256]]'))
for crops, ext in CFG:
    geo = fused.PathGeometry(dim=2, method=1, step_number=0.25, mip_level=0, extent=(ext, ext), num_crops=crops,
                             noise_mode=2, noise_seed=7, noise_offset=1, split_bf16=True)
    org = torch.randint(0, S - ext + 1, (crops, 2), dtype=torch.int32, device=dev)
    target = torch.rand(crops * ext * ext, 3, device=dev)
    ts = []
    for i in range(30):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fused.fused_forward_backward(geo, fp[0].detach(), fp[1].detach(), org, params, target)
        e1.record(); torch.cuda.synchronize()
        if i >= 10: ts.append(e0.elapsed_time(e1))
    print(f"crops {crops:2d} x {ext}^2: step (fused + reduce) median {np.median(ts)*1e3:7.1f} us  min {np.min(ts)*1e3:7.1f} us  -> {crops*ext*ext/np.median(ts)/1e3:7.1f} Mpix/s")
