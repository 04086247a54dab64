import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests.test_gpu_tail import _fit
from neural_image_compression_v2_amd import _lib, fused
dev = torch.device("cuda:0")
res = {}
for mode in ("separate", "tail"):
    masters, _, params, opt = _fit(dev, 2, 1, 3, torch.float32, seed=8)
    geo = fused.PathGeometry(dim=2, method=1, step_number=0.25, mip_level=0, extent=(64, 64), num_crops=1, noise_mode=_lib.NIC_NOISE_KERNEL, noise_seed=1,
                             noise_offset=2, split_bf16=True, flags=_lib.NIC_FLAG_ORIGINS_ALIGNED)
    target = torch.rand(64 * 64, 3, generator=torch.Generator().manual_seed(3)).to(dev)
    tail = (lambda gg0, gg1, gm: opt.step_tail([(masters[0], gg0), (masters[1], gg1)], list(zip(params, gm)))) if mode == "tail" else None
    p0 = [p.detach().clone() for p in params]
    out = fused.fused_forward_backward(geo, masters[0], masters[1], [(8, 16)], params, target, tail=tail)
    g_before = [t.clone() for t in out.grad_mlp]
    masters[0].grad, masters[1].grad = out.grad_g0, out.grad_g1
    for p, gq in zip(params, out.grad_mlp):
        p.grad = gq
    opt.step()
    torch.cuda.synchronize()
    res[mode] = (g_before, [p.detach().clone() for p in params], [opt.state[p]["exp_avg"].clone() for p in params], [opt.state[p]["exp_avg_sq"].clone() for p in params], p0,
                 [masters[0].detach().clone(), masters[1].detach().clone()])
for k, what in enumerate(("grads", "params", "m", "v", "p0", "grids")):
    for i, (x, y) in enumerate(zip(res["separate"][k], res["tail"][k])):
        d = (x - y).abs()
        print(what, i, "maxdiff", float(d.max()), "n differ", int((d > 0).sum()), "of", d.numel())
# expected update from torch formula
g, p0 = res["tail"][0][0], res["tail"][4][0]
m = 0.1 * g; v = 0.001 * g * g
print("upd tail", float((res["tail"][1][0] - p0).abs().max()), "sep", float((res["separate"][1][0] - p0).abs().max()))
import numpy as np
ps, pt = res["separate"][1][0].flatten().cpu().numpy(), res["tail"][1][0].flatten().cpu().numpy()
g = res["tail"][0][0].flatten().cpu().numpy(); p0 = res["tail"][4][0].flatten().cpu().numpy()
idx = np.nonzero(ps != pt)[0][:8]
f32 = np.float32
for i in idx:
    gi = f32(g[i]); m = f32(f32(0) + f32(f32(gi - f32(0)) * f32(1.0 - 0.9))); v = f32(f32(f32(0) * f32(0.999)) + f32(f32(f32(1.0 - 0.999) * gi) * gi))
    bc1 = 1.0 - 0.9; bc2 = 1.0 - 0.999
    ss = f32(0.005 / bc1); b2s = f32(np.sqrt(bc2))
    den = f32(f32(np.sqrt(v, dtype=np.float32) / b2s) + f32(1e-8))
    x = f32(p0[i] - f32(ss * f32(m / den)))
    print(i, "p0", p0[i], "g", gi, "sep", ps[i], "tail", pt[i], "numpy op-by-op", x, "sep==np", ps[i] == x, "tail==np", pt[i] == x)
us, ut = (ps.astype(np.float64) - p0), (pt.astype(np.float64) - p0)
r = ut / us
print("ratio of updates tail/sep: min", r.min(), "max", r.max(), "mean", r.mean(), "median", np.median(r))
big = np.abs(us) > 0.004
print("ratio on full-size updates: min", r[big].min(), "max", r[big].max(), "mean-1", r[big].mean() - 1)
g = g.astype(np.float32); p0 = p0.astype(np.float32)
f = np.float32
def fma(a, b, c): return (a.astype(np.float64) * np.float64(b) + np.asarray(c, dtype=np.float64)).astype(np.float32)
omb1, b2, omb2, eps = f(1.0 - 0.9), f(0.999), f(1.0 - 0.999), f(1e-8)
ss, b2s = f(0.005 / (1.0 - 0.9)), f(np.sqrt(1.0 - 0.999))
m = (g * omb1).astype(np.float32)
v = ((omb2 * g).astype(np.float32) * g).astype(np.float32)
sq = np.sqrt(v.astype(np.float64)).astype(np.float32)
cands = {}
den = ((sq / b2s).astype(np.float32) + eps).astype(np.float32)
q = (m / den).astype(np.float32)
cands["op-by-op"] = (p0 - (ss * q).astype(np.float32)).astype(np.float32)
cands["fma last"] = fma(q, -ss, p0)
den2 = fma(sq, 1.0, 0) ; den2 = ((sq.astype(np.float64) / np.float64(b2s)) + np.float64(eps)).astype(np.float32)
q2 = (m / den2).astype(np.float32)
cands["den unrounded"] = (p0 - (ss * q2).astype(np.float32)).astype(np.float32)
q3 = (m.astype(np.float64) / den.astype(np.float64))
cands["q unrounded"] = (p0 - (np.float64(ss) * q3).astype(np.float32)).astype(np.float32)
cands["q unrounded + fma"] = (p0.astype(np.float64) - np.float64(ss) * q3).astype(np.float32)
m2 = (g.astype(np.float64) * np.float64(omb1))
cands["m unrounded"] = (p0 - (ss * (m2 / den.astype(np.float64)).astype(np.float32)).astype(np.float32)).astype(np.float32)
for k, c in cands.items():
    print(f"{k:20s} == sep: {int((c == ps).sum())} / {c.size}   == tail: {int((c == pt).sum())} / {c.size}")
