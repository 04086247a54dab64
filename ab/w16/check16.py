"""debug harness: the 8-wave / 16-sample training kernel against the fp32 kernel (and the CPU oracle) on small cases, per-tensor
errors with the position of the worst element"""
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
    return float(d.max() / (b.abs().max() + 1e-300)), idx, float(a.reshape(-1)[i]) if a.dim() else float(a), float(b.reshape(-1)[i]) if b.dim() else float(b)


def run(name, extent, origins, noise_kind="none", tri=True, mip=0, fl=0, base=64, passes=1, oracle=True):
    g = torch.Generator().manual_seed(9)
    fp, _ = O.create_pyramid(base, 12, 8, dim=2, no_mip=(mip == 0), generator=g)
    g0, g1 = fp[2 * fl].detach(), fp[2 * fl + 1].detach()
    step = O.step_number_of(mip, fl)
    mlp = O.init_mlp(73, 64, generator=g)
    n = len(origins) * int(np.prod(extent)) * passes
    target = torch.rand(n, 3, generator=g)
    kw, noise, nd = {}, None, None
    if noise_kind == "kernel":
        noise = O.kernel_noise(n, 73, 8, seed=5, offset=6)
        kw = dict(noise_mode=_lib.NIC_NOISE_KERNEL, noise_seed=5, noise_offset=6)
    elif noise_kind == "tensor":
        noise = (torch.rand(n, 73, generator=g) - 0.5) / 256
        kw = dict(noise_mode=_lib.NIC_NOISE_TENSOR)
        nd = noise.to(dev)
    params = [q.to(dev) for q in mlp.tensors()]
    outs = {}
    for tag, sp, t32 in (("f32", False, False), ("t32", True, True), ("t16", True, False)):
        geo = fused.PathGeometry(dim=2, method=1, step_number=step, mip_level=mip, extent=extent, num_crops=len(origins), use_tri_pe=tri,
                                 split_bf16=sp, split_tile32=t32, passes=passes, **kw)
        outs[tag] = fused.fused_forward_backward(geo, g0.to(dev), g1.to(dev), origins, params, target.to(dev), nd, want_y=True)
        torch.cuda.synchronize()
    ref = outs["f32"]
    print(f"== {name}: extent {extent} origins {origins} noise {noise_kind} mip {mip}")
    for tag in ("t32", "t16"):
        o = outs[tag]
        items = [("y", o.y, ref.y), ("loss", o.loss, ref.loss), ("G0", o.grad_g0, ref.grad_g0), ("G1", o.grad_g1, ref.grad_g1)] + \
                [(nm, a, b) for nm, a, b in zip(["W1", "b1", "W2", "b2", "W3", "b3"], o.grad_mlp, ref.grad_mlp)]
        line = []
        for nm, a, b in items:
            e, idx, va, vb = err(a, b)
            line.append(f"{nm} {e:.1e}" + (f"@{tuple(int(v) for v in idx)}[{va:.4g} vs {vb:.4g}]" if e > 1e-4 else ""))
        print(f"  {tag}: " + "  ".join(line))
    if oracle and passes == 1:
        r = O.forward_backward(g0, g1, mlp, origins, extent, step, mip, target, noise, 6, use_tri_pe=tri)
        o = outs["t16"]
        items = [("y", o.y, r.y), ("loss", o.loss, r.loss), ("G0", o.grad_g0, r.grad_g0), ("G1", o.grad_g1, r.grad_g1)] + \
                [(nm, a, b) for nm, a, b in zip(["W1", "b1", "W2", "b2", "W3", "b3"], o.grad_mlp, r.grad_mlp)]
        print("  t16 vs oracle: " + "  ".join(f"{nm} {err(a, b)[0]:.1e}" for nm, a, b in items))


run("one tile aligned", (64, 4), [(0, 0)])
run("aligned multi", (64, 64), [(0, 0), (64, 128)], "kernel")
run("unaligned", (37, 21), [(3, 5), (200, 100)], "kernel")
run("tensor noise", (40, 24), [(3, 5), (20, 0)], "tensor", tri=False)
run("default shape", (256, 256), [(0, 0), (0, 0)], "kernel")
run("mip1", (40, 24), [(3, 5), (50, 30)], "kernel", mip=1)
run("mip2", (20, 24), [(3, 5), (20, 7)], "tensor", mip=2)
run("passes", (24, 40), [(0, 8), (100, 60)], "kernel", passes=3, oracle=False)
