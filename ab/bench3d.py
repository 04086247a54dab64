"""quick 3D throughput check (method 3 and 4): one pass over a 128^3 volume with 64-level grids"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_image_compression_v2_amd import _lib, fused, fp_def
from neural_image_compression_v2_amd.image_compression import ColorDecoder
dev = torch.device("cuda:0")
S = 128
for method, cin in ((3, 127), (4, 79)):
    torch.manual_seed(0)
    fp, _ = fp_def.create_pyramid_3d(S // 4, 12, 8, dev, torch.float32, True)
    dec = ColorDecoder(cin, 64).to(dev)
    params = [p.detach() for p in dec.linear_params()]
    target = torch.rand(S ** 3, 3, device=dev)
    org = [[0, 0, 0]]                                        # host origins: the wrapper sees they are cell-aligned
    SPLIT_TRAIN = os.environ.get("SPLIT3D", "1") == "1"
    def step(i):
        geo = fused.PathGeometry(dim=3, method=method, step_number=0.25, mip_level=0, extent=(S, S, S), num_crops=1,
                                 noise_mode=_lib.NIC_NOISE_KERNEL, noise_seed=7, noise_offset=i, split_bf16=SPLIT_TRAIN)
        return fused.fused_forward_backward(geo, fp[0].detach(), fp[1].detach(), org, params, target)
    for i in range(3): step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(10): out = step(3 + i)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"3D method {method}: {S}^3 voxels fwd+bwd {dt*1e3:.2f} ms -> {S**3/dt/1e6:.1f} Mvox/s, loss {float(out.loss):.5f}")
    for split in (False, True):
        gd = fused.PathGeometry(dim=3, method=method, step_number=0.25, mip_level=0, extent=(S, S, S), num_crops=1, split_bf16=split)
        for i in range(3): fused.fused_forward(gd, fp[0].detach(), fp[1].detach(), org, params)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(10): y = fused.fused_forward(gd, fp[0].detach(), fp[1].detach(), org, params)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print(f"   decode ({'split' if split else 'fp32'}): {dt*1e3:.2f} ms -> {S**3/dt/1e6:.0f} Mvox/s")
