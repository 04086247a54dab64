"""diagnostic: where do the periodic ~90 ms host stalls of the training loop come from"""
import os, sys, time, math, random, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_image_compression_v2_amd.image_compression import ImageCompression
from neural_image_compression_v2_amd.var2 import Settings
dev = torch.device("cuda:0")
def run(tag, n=400):
    cfg = Settings(IMAGE_SIZE=512, NUM_EPOCHS=2000, TF_NO_MIP=True, NUM_CROPS=8)
    S = cfg.IMAGE_SIZE
    u = torch.linspace(0, 1, S)
    img = torch.stack([0.5 + 0.25 * torch.sin(2 * math.pi * (c + 1) * u)[:, None] * torch.cos(2 * math.pi * (c + 2) * u)[None, :] for c in range(3)])
    ic = ImageCompression(cfg, dev, seed=0)
    ic.set_images([torch.round(img.clamp(0, 1) * 255).to(torch.uint8)])
    torch.manual_seed(1); random.seed(1)
    for e in range(30): ic.train_step(ic.feature_pyramid, e)
    torch.cuda.synchronize()
    ts = []
    t0 = time.perf_counter()
    for e in range(30, 30 + n):
        a = time.perf_counter(); ic.train_step(ic.feature_pyramid, e); ts.append(time.perf_counter() - a)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    ts = sorted(ts)
    print(f"{tag}: {1e6*(t1-t0)/n:.1f} us/step; host per step median {1e6*ts[n//2]:.1f} us, p99 {1e6*ts[int(n*0.99)]:.1f}, max {1e6*ts[-1]:.1f}, steps > 5 ms: {sum(t > 5e-3 for t in ts)}")
run("StepPlan")

os.environ["NIC_NO_PLAN"] = "1"; run("no StepPlan")
