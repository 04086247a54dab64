"""debug harness for the depth-generic kernel (fused_mlpn.hpp): n_linear = 3 through NIC_FLAG_MLPN against the fp32 kernel, n_linear = 5
against the CPU oracle; training and decode"""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neural_image_compression_v2_amd import _lib, fused
from oracle import nic_oracle as O

dev = torch.device("cuda:0")


def err(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    d = (a - b).abs()
    i = int(d.argmax())
    idx = np.unravel_index(i, tuple(a.shape)) if a.dim() else ()
    return float(d.max() / (b.abs().max() + 1e-300)), tuple(int(v) for v in idx)


def run(nl, extent, origins, noise_kind="kernel", tri=True, mip=0, passes=1):
    g = torch.Generator().manual_seed(9)
    fp, _ = O.create_pyramid(64, 12, 8, dim=2, no_mip=(mip == 0), generator=g)
    g0, g1 = fp[0].detach(), fp[1].detach()
    step = O.step_number_of(mip, 0)
    mlp = O.init_mlp(73, 64, generator=g, n_linear=nl)
    n = len(origins) * int(np.prod(extent)) * passes
    target = torch.rand(n, 3, generator=g)
    kw, noise = {}, None
    if noise_kind == "kernel":
        noise = O.kernel_noise(n, 73, 8, seed=5, offset=6)
        kw = dict(noise_mode=_lib.NIC_NOISE_KERNEL, noise_seed=5, noise_offset=6)
    params = [q.to(dev) for q in mlp.tensors()]
    geo = fused.PathGeometry(dim=2, method=1, step_number=step, mip_level=mip, extent=extent, num_crops=len(origins), use_tri_pe=tri,
                             split_bf16=True, mlpn=True, passes=passes, **kw)
    out = fused.fused_forward_backward(geo, g0.to(dev), g1.to(dev), origins, params, target.to(dev), want_y=True)
    yi = fused.fused_forward(geo, g0.to(dev), g1.to(dev), origins, params) if passes == 1 else out.y
    torch.cuda.synchronize()
    names = [f"{k}{i + 1}" for i in range(nl) for k in ("W", "b")]
    if passes == 1:
        r = O.forward_backward(g0, g1, mlp, origins, extent, step, mip, target, noise, 6, use_tri_pe=tri)
        items = [("y", out.y, r.y), ("yinf", yi, r.y), ("loss", out.loss, r.loss), ("G0", out.grad_g0, r.grad_g0), ("G1", out.grad_g1, r.grad_g1)] + \
                [(nm, a, b) for nm, a, b in zip(names, out.grad_mlp, r.grad_mlp)]
        print(f"NL={nl} {extent} {origins} mip{mip} vs oracle: " + "  ".join(f"{nm} {err(a, b)[0]:.1e}" + (f"@{err(a, b)[1]}" if err(a, b)[0] > 1e-4 else "") for nm, a, b in items))
    if nl == 3:
        geo3 = fused.PathGeometry(dim=2, method=1, step_number=step, mip_level=mip, extent=extent, num_crops=len(origins), use_tri_pe=tri, passes=passes, **kw)
        ref = fused.fused_forward_backward(geo3, g0.to(dev), g1.to(dev), origins, params, target.to(dev), want_y=True)
        items = [("y", out.y, ref.y), ("loss", out.loss, ref.loss), ("G0", out.grad_g0, ref.grad_g0), ("G1", out.grad_g1, ref.grad_g1)] + \
                [(nm, a, b) for nm, a, b in zip(names, out.grad_mlp, ref.grad_mlp)]
        print(f"NL=3 {extent} p{passes} vs fp32 kernel: " + "  ".join(f"{nm} {err(a, b)[0]:.1e}" + (f"@{err(a, b)[1]}" if err(a, b)[0] > 1e-4 else "") for nm, a, b in items))


for nl in (3, 5):
    run(nl, (64, 4), [(0, 0)], "none")
    run(nl, (64, 64), [(0, 0), (64, 128)])
    run(nl, (37, 21), [(3, 5), (200, 100)], tri=False)
    run(nl, (256, 256), [(0, 0), (0, 0)])
    run(nl, (40, 24), [(3, 5), (50, 30)], mip=1)
    run(nl, (24, 40), [(0, 8), (100, 60)], passes=3)
