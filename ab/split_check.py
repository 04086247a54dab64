"""split-bf16 mode vs the fp32 kernel: error levels and 4K timing (diagnostic)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_image_compression_v2_amd import _lib, fused, fp_def
from neural_image_compression_v2_amd.image_compression import ColorDecoder
dev = torch.device("cuda:0")
def rel(a, b): return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-300))
torch.manual_seed(0)
for (H, W, crops, org) in ((64, 96, 2, [[3, 5], [100, 60]]), (2160, 3840, 1, [[0, 0]])):
    fp, _ = fp_def.create_pyramid((max(H, 256) // 4 + 16, max(W, 256) // 4 + 16), 12, 8, dev, torch.float32, True)
    dec = ColorDecoder(73, 64).to(dev)
    params = [p.detach() for p in dec.linear_params()]
    target = torch.rand(crops * H * W, 3, device=dev)
    kw = dict(dim=2, method=1, step_number=0.25, mip_level=0, extent=(H, W), num_crops=crops, noise_mode=_lib.NIC_NOISE_KERNEL, noise_seed=7, noise_offset=1)
    g32 = fused.PathGeometry(**kw)
    gsp = fused.PathGeometry(**kw, flags=_lib.NIC_FLAG_SPLIT_BF16)
    a = fused.fused_forward_backward(g32, fp[0].detach(), fp[1].detach(), org, params, target, want_y=True)
    b = fused.fused_forward_backward(gsp, fp[0].detach(), fp[1].detach(), org, params, target, want_y=True)
    print(f"{H}x{W}: y {rel(b.y, a.y):.2e} loss {rel(b.loss, a.loss):.2e} gG0 {rel(b.grad_g0, a.grad_g0):.2e} gG1 {rel(b.grad_g1, a.grad_g1):.2e} "
          + " ".join(f"{rel(x, y):.1e}" for x, y in zip(b.grad_mlp, a.grad_mlp)))
    if H > 1000:
        for name, g in (("fp32", g32), ("split", gsp)):
            for _ in range(3): fused.fused_forward_backward(g, fp[0].detach(), fp[1].detach(), org, params, target)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): fused.fused_forward_backward(g, fp[0].detach(), fp[1].detach(), org, params, target)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
            print(f"  {name}: {dt*1e3:.3f} ms -> {H*W/dt/1e6:.0f} Mpix/s")
