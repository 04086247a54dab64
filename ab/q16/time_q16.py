"""kernel-only timings of the 4K step: split (train16 / mlpn) against the plain-bf16 q16 kernels, fp32 and bf16 grid storage.
   python ab/q16/time_q16.py [launches]"""
import os, sys, json
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from neural_image_compression_v2_amd import _lib, fp_def, fused
from neural_image_compression_v2_amd.image_compression import ColorDecoder

H, W = 2160, 3840
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
which = sys.argv[2].split(",") if len(sys.argv) > 2 else None
torch.manual_seed(0)
fp, _ = fp_def.create_pyramid((H // 4, W // 4), 12, 8, dev, torch.float32, True)
g0, g1 = fp[0].detach(), fp[1].detach()
target = torch.rand(H * W, 3, device=dev)
org = torch.zeros(1, 2, dtype=torch.int32, device=dev)
res = {}
for name, nl, kw, gdt in [("split_nl3", 3, dict(split_bf16=True), torch.float32), ("bf16_nl3", 3, dict(bf16=True), torch.float32),
                          ("bf16_nl3_g16", 3, dict(bf16=True), torch.bfloat16), ("split_nl5", 5, dict(split_bf16=True), torch.float32),
                          ("bf16_nl5", 5, dict(bf16=True), torch.float32), ("bf16_nl5_g16", 5, dict(bf16=True), torch.bfloat16),
                          ("fp16_nl3", 3, dict(fp16=True), torch.float32), ("fp16_nl5", 5, dict(fp16=True), torch.float32), ("fp16_nl5_g16", 5, dict(fp16=True), torch.float16)]:
    if which and name not in which:
        continue
    dec = ColorDecoder(73, 64, nl).to(dev)
    params = [p.detach() for p in dec.linear_params()]
    a, b = g0.to(gdt), g1.to(gdt)
    flat = None
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for j in range(n + 5):
        geo = fused.PathGeometry(dim=2, method=1, step_number=0.25, mip_level=0, extent=(H, W), num_crops=1, noise_mode=_lib.NIC_NOISE_KERNEL, use_tri_pe=os.environ.get("TRI", "1") != "0",
                                 noise_seed=7, noise_offset=j, flags=_lib.NIC_FLAG_ORIGINS_ALIGNED, **kw)
        o = fused.fused_forward_backward(geo, a, b, org, params, target, flat=flat, events=evs[j - 5] if j >= 5 else None)
        flat = o.flat
    torch.cuda.synchronize()
    t = np.array([x.elapsed_time(y) for x, y in evs])
    res[name] = {"median_ms": round(float(np.median(t)), 4), "min_ms": round(float(t.min()), 4), "mpix_s": round(H * W / float(np.median(t)) / 1e3, 1), "loss": float(o.loss)}
    print(name, res[name], flush=True)
# 3D: a 128^3 volume, methods 3 and 4: the 32-sample kernels (chained split products) against the plain-bf16 quarter kernels
S3 = 128
fp3, _ = fp_def.create_pyramid_3d(S3 // 4, 12, 8, dev, torch.float32, True)
t3 = torch.rand(S3 ** 3, 3, device=dev)
org3 = torch.zeros(1, 3, dtype=torch.int32, device=dev)
for method in (3, 4):
    cin = 127 if method == 3 else 79
    dec = ColorDecoder(cin, 64, 3).to(dev)
    params = [p.detach() for p in dec.linear_params()]
    for name, kw, gdt in [(f"m{method}_split", dict(split_bf16=True), torch.float32), (f"m{method}_bf16", dict(bf16=True), torch.float32),
                          (f"m{method}_bf16_g16", dict(bf16=True), torch.bfloat16), (f"m{method}_fp16", dict(fp16=True), torch.float32)]:
        if which and name not in which:
            continue
        a, b = fp3[0].detach().to(gdt), fp3[1].detach().to(gdt)
        flat = None
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for j in range(n + 5):
            geo = fused.PathGeometry(dim=3, method=method, step_number=0.25, mip_level=0, extent=(S3, S3, S3), num_crops=1, noise_mode=_lib.NIC_NOISE_KERNEL,
                                     noise_seed=7, noise_offset=j, flags=_lib.NIC_FLAG_ORIGINS_ALIGNED, **kw)
            o = fused.fused_forward_backward(geo, a, b, org3, params, t3, flat=flat, events=evs[j - 5] if j >= 5 else None)
            flat = o.flat
        torch.cuda.synchronize()
        t = np.array([x.elapsed_time(y) for x, y in evs])
        res[name] = {"median_ms": round(float(np.median(t)), 4), "min_ms": round(float(t.min()), 4), "mvox_s": round(S3 ** 3 / float(np.median(t)) / 1e3, 1), "loss": float(o.loss)}
        print(name, res[name], flush=True)
print(json.dumps(res))
