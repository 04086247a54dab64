"""debug: fused multi-level gradients against the fp32 oracle on small crops (summary per grid + the worst node)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import nic_oracle as O
from neural_image_compression_v2_amd import _lib, fused
from neural_image_compression_v2_amd.multilevel import level_nodes
torch.set_printoptions(linewidth=220, precision=4, sci_mode=False)
dev = torch.device("cuda:0")
L, C, NL, P = 2, 4, 3, 6
size = (128, 96)
g = torch.Generator().manual_seed(11)
fp = []
for l in range(L):
    for nodes in level_nodes(size, l):
        fp.append(torch.rand(C, nodes[1], nodes[0], generator=g) - 0.498)
cin = L * (5 * C + 2 * P) + 1
mlp = O.init_mlp(cin, 64, torch.Generator().manual_seed(5), n_linear=NL)
for ext, origins in [((48, 4), [(0, 16)]), ((48, 16), [(0, 16)]), ((48, 8), [(0, 12)]), ((48, 20), [(0, 0)]), ((48, 4), [(0, 0), (0, 16)]), ((48, 4), [(0, 0), (0, 4), (0, 8), (0, 12), (0, 16)])]:
    n = len(origins) * ext[0] * ext[1]
    target = torch.rand(n, 3, generator=g)
    ref = O.multilevel_forward_backward(fp, mlp, origins, ext, target, None, P, True)
    geo = fused.PathGeometry(dim=2, method=1, step_number=0.25, mip_level=0, extent=ext, num_crops=len(origins), channels=C, pe_channels=P)
    out = fused.fused_ml_forward_backward(geo, [t.to(dev) for t in fp], origins, [t.to(dev) for t in mlp.tensors()], target.to(dev), want_y=True)
    print("== extent", ext, "origins", origins, "y err", float((out.y.cpu() - ref[0]).abs().max()))
    for i, (a, b) in enumerate(zip(out.grad_fp, ref[2])):
        a = a.cpu()
        d = (a - b).abs()
        k = int(d.argmax())
        idx = tuple(int(v) for v in torch.unravel_index(torch.tensor(k), d.shape))
        print(f"  grid {i} {tuple(a.shape)} sum out {float(a.sum()):.3e} ref {float(b.sum()):.3e} max|diff| {float(d.max()):.2e} max|ref| {float(b.abs().max()):.2e} worst {idx} out {float(a[idx]):.3e} ref {float(b[idx]):.3e}")
