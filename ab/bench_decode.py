"""decode throughput of a 4K image: fp32 grids vs stored uint8 grids (one launch each), float and byte output"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_image_compression_v2_amd import fused, fp_def, models
from neural_image_compression_v2_amd.image_compression import ColorDecoder
H, W = 2160, 3840
dev = torch.device("cuda:0")
torch.manual_seed(0)
fp, _ = fp_def.create_pyramid((H // 4, W // 4), 12, 8, dev, torch.float32, True)
fq = fp_def.fp_all_quantize([g.detach() for g in fp], 8)
fu = fp_def.fp_savable([g.detach() for g in fp], 8, torch.uint8)
dec = ColorDecoder(73, 64).to(dev)
params = [p.detach() for p in dec.linear_params()]
geo = fused.PathGeometry(dim=2, method=1, step_number=0.25, mip_level=0, extent=(H, W), num_crops=1)
org = [[0, 0]]
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
a = timeit(lambda: fused.fused_forward(geo, fq[0], fq[1], org, params))
b = timeit(lambda: fused.fused_forward_u8(geo, fu[0], fu[1], org, params))
c = timeit(lambda: fused.fused_forward_u8(geo, fu[0], fu[1], org, params, out="uint8"))
gs = fused.PathGeometry(dim=2, method=1, step_number=0.25, mip_level=0, extent=(H, W), num_crops=1, split_bf16=True)
a2 = timeit(lambda: fused.fused_forward(gs, fq[0], fq[1], org, params))
b2 = timeit(lambda: fused.fused_forward_u8(gs, fu[0], fu[1], org, params))
ya, yb = fused.fused_forward(geo, fq[0], fq[1], org, params), fused.fused_forward(gs, fq[0], fq[1], org, params)
print(f"split-bf16 decode 4K: fp32 grids {a2*1e3:.3f} ms ({H*W/a2/1e6:.0f} Mpix/s) | uint8 grids {b2*1e3:.3f} ms ({H*W/b2/1e6:.0f} Mpix/s); max |y_split - y_f32| = {float((ya - yb).abs().max()):.2e}, "
      f"bytes differing: {int((torch.floor(ya * 255 + 0.5) != torch.floor(yb * 255 + 0.5)).sum())} of {ya.numel()}")
print(f"decode 4K: fp32 grids {a*1e3:.3f} ms ({H*W/a/1e6:.0f} Mpix/s) | uint8 grids {b*1e3:.3f} ms ({H*W/b/1e6:.0f} Mpix/s) | uint8 grids -> bytes {c*1e3:.3f} ms ({H*W/c/1e6:.0f} Mpix/s)")
assert torch.equal(fused.fused_forward(geo, fq[0], fq[1], org, params), fused.fused_forward_u8(geo, fu[0], fu[1], org, params))
