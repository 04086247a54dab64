"""fwd + bwd time of the general decoder kernels (nic_decoder_general_*) on [n, Cin] inputs: python ab/time_general.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_image_compression_v2_amd.image_compression import ColorDecoder
dev = torch.device("cuda:0")
for cin, H, NL, n in [(73, 128, 3, 524288), (73, 96, 3, 524288), (361, 64, 5, 552960), (127, 256, 5, 262144), (73, 64, 4, 524288)]:
    dec = ColorDecoder(cin, H, NL).to(dev)
    x = (torch.rand(n, cin, device=dev) - 0.5).requires_grad_(True)
    dy = torch.randn(n, 3, device=dev)
    for it in range(6):
        if it == 2:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        y = dec(x)
        y.backward(dy)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 4
    flop = 6 * n * (cin * H + (NL - 2) * H * H + 3 * H)
    print(f"Cin {cin} H {H} NL {NL} n {n}: {dt * 1e3:.2f} ms fwd+bwd, {flop / dt / 1e12:.1f} TFLOP/s (fp32 matrix peak 157)", flush=True)
