"""kernel-trace-free timing of small steps (HIP events around the fused call): the reference's 3D sweep shape (8 x 32^3 crops of a 64^3 volume, methods 3 / 4)
and its 2D default step (8 x 256^2 crops of a 512^2 image), plain-bf16 and split products.   python ab/q16/time_small.py [launches]"""
import os, sys, json
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from neural_image_compression_v2_amd import _lib, fp_def, fused
from neural_image_compression_v2_amd.image_compression import ColorDecoder
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
torch.manual_seed(0)
rs = np.random.RandomState(0)
for name, dim, method, size, crop in [("sweep_m3", 3, 3, 64, 32), ("sweep_m4", 3, 4, 64, 32), ("default_2d", 2, 1, 512, 256)]:
    mk = fp_def.create_pyramid if dim == 2 else fp_def.create_pyramid_3d
    fp, _ = mk(size // 4, 12, 8, dev, torch.float32, True)
    cin = {1: 73, 3: 127, 4: 79}[method]
    dec = ColorDecoder(cin, 64, 3).to(dev)
    params = [p.detach() for p in dec.linear_params()]
    img = torch.randint(0, 256, (3,) + (size,) * dim, dtype=torch.uint8, device=dev)
    tgt = fused.TargetImage(img, 255.0 if dim == 2 else 256.0)
    for prec in ("bf16", "split"):
        flat = None
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for j in range(n + 10):
            org = [tuple(int(rs.randint(0, size - crop + 1)) for _ in range(dim)) for _ in range(8)]
            geo = fused.PathGeometry(dim=dim, method=method, step_number=0.25, mip_level=0, extent=(crop,) * dim, num_crops=8, noise_mode=_lib.NIC_NOISE_KERNEL,
                                     noise_seed=7, noise_offset=j, split_bf16=prec == "split", bf16=prec == "bf16")
            o = fused.fused_forward_backward(geo, fp[0].detach(), fp[1].detach(), org, params, tgt, flat=flat, events=evs[j - 10] if j >= 10 else None)
            flat = o.flat
        torch.cuda.synchronize()
        t = np.array([x.elapsed_time(y) for x, y in evs]) * 1e3
        print(f"{name} {prec}: kernel + reduce median {np.median(t):.1f} us (min {t.min():.1f})", flush=True)
