"""diagnostic: where does the host stall in a short timed loop of StepPlan steps (bench_workloads._run_volume shape, 128^3 method 3)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
dev = torch.device("cuda:0")
fit = bench.Fit(dev, 3, 3, grid_base=(32, 32, 32), precision="bf16")
ext = (128, 128, 128)
target = torch.rand(int(np.prod(ext)), 3).to(dev)
org = [[0, 0, 0]]
use_ev = os.environ.get("EV", "1") == "1"
def step(i, ev=None):
    out = fit.fwd_bwd(fit.geometry(i, ext, aligned=True), org, target, ev, adam=(i, 45))
    fit.adam(out, i, 45)
for i in range(5):
    step(i)
torch.cuda.synchronize()
evs = [bench.KernelEvents(fit.lib) for _ in range(40)]
ts = [time.perf_counter()]
for i in range(40):
    step(5 + i, evs[i] if use_ev else None)
    ts.append(time.perf_counter())
torch.cuda.synchronize()
ts.append(time.perf_counter())
d = np.diff(ts) * 1e3
print("host ms per step:", " ".join(f"{x:.2f}" for x in d[:-1]), "| final sync", f"{d[-1]:.2f}", "| total", f"{(ts[-1]-ts[0])*1e3:.1f} ms")
