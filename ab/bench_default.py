"""the reference's own default shape: IMAGE_SIZE 512, 8 crops of 256 x 256 per step (image_compression.py / var2.py defaults), through the
product's host loop (random origins from the host RNGs, targets from the resident uint8 image, fused step, one-launch Adam)"""
import os, sys, time, math, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_image_compression_v2_amd.image_compression import ImageCompression
from neural_image_compression_v2_amd.var2 import Settings
dev = torch.device("cuda:0")
for split in (True, False):
    cfg = Settings(IMAGE_SIZE=512, NUM_EPOCHS=300, TF_NO_MIP=True, TF_SPLIT_BF16=split)
    S = cfg.IMAGE_SIZE
    u = torch.linspace(0, 1, S)
    img = torch.stack([0.5 + 0.25 * torch.sin(2 * math.pi * (c + 1) * u)[:, None] * torch.cos(2 * math.pi * (c + 2) * u)[None, :] for c in range(3)])
    ic = ImageCompression(cfg, dev, seed=0)
    ic.set_images([torch.round(img.clamp(0, 1) * 255).to(torch.uint8)])
    torch.manual_seed(1); random.seed(1)
    for e in range(20): ic.train_step(ic.feature_pyramid, e)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 200
    for e in range(20, 20 + n): ic.train_step(ic.feature_pyramid, e)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    px = cfg.NUM_CROPS * 256 * 256
    print(f"default shape ({'split-bf16' if split else 'fp32'} products): {dt*1e3:.3f} ms / step = {px/dt/1e6:.0f} Mpix/s, PSNR after {20+n} steps {float(ic.psnr(ic.feature_pyramid)):.2f} dB")
