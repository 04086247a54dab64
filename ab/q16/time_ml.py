"""kernel-only timing of the fused multi-level step at 4K (3840 x 2160, one launch): python ab/q16/time_ml.py [launches] [L,C,NL ...]"""
import os, sys, json
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from neural_image_compression_v2_amd import _lib, fused
from neural_image_compression_v2_amd.multilevel import level_nodes
from neural_image_compression_v2_amd.image_compression import ColorDecoder

H, W = 2160, 3840
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cases = [tuple(int(v) for v in a.split(",")) for a in sys.argv[2:]] or [(5, 4, 3), (3, 4, 5), (2, 4, 5), (3, 12, 3), (2, 12, 3)]
torch.manual_seed(0)
target = torch.rand(H * W, 3, device=dev)
for L, C, NL in cases:
    fp = []
    for l in range(L):
        for nodes in level_nodes((H, W), l):
            fp.append(torch.rand(C, nodes[1], nodes[0], device=dev) - 0.498)
    cin = L * (5 * C + 12) + 1
    dec = ColorDecoder(cin, 64, NL).to(dev)
    params = [p.detach() for p in dec.linear_params()]
    grads = [torch.zeros_like(g) for g in fp]
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for j in range(n + 3):
        geo = fused.PathGeometry(dim=2, method=1, step_number=0.25, mip_level=0, extent=(H, W), num_crops=1, channels=C, noise_mode=_lib.NIC_NOISE_KERNEL,
                                 noise_seed=7, noise_offset=j, flags=_lib.NIC_FLAG_ORIGINS_ALIGNED)
        o = fused.fused_ml_forward_backward(geo, fp, [[0, 0]], params, target, grads=grads, events=evs[j - 3] if j >= 3 else None)
    torch.cuda.synchronize()
    t = np.array([a.elapsed_time(b) for a, b in evs])
    print(f"ml L{L} C{C} NL{NL} cin {cin}: median {np.median(t):.3f} ms  min {t.min():.3f}  = {H * W / np.median(t) / 1e3:.0f} Mpix/s  loss {float(o.loss):.5f}", flush=True)
