"""`python bench.py --workload NAME`: the BASELINE.json configurations beyond the headline 4K fit, one GPU, same JSON line shape
(Mvoxels/s or Mpixels/s of one training step = fused fwd + bwd, plus Adam + clamp, HIP-event kernel times, the
algorithmic-bytes roofline of the survey with the matrix-pipe figures beside it).  The driver runs the default workload only; these
records are kept under profiles/.

    lut33    config 3: a 33^3 colour LUT (RGB -> RGB) as one crop, 3D method 3 (the reference's permuted trilinear weights), grids
             ceil(33/4)+1 = 10 and 6 nodes per axis; N = 35 937 per step (launch-latency regime)
    vol64    the 64^3 volume the reference's own sweeps use (method 3 and 4)
    vol128   a 128^3 volume, method 3 and 4
    slab     config 4, one rank's slab of the 1920 x 1080 x 64 video field (the whole field, any --gpus: bench.py --workload video) (x <-> T, y <-> H, z <-> W; grids [12,481,271,17] +
             [12,241,136,9] at full size): z in [240 r, 240 r + 240), 16.6 Mvox per step, method 4 (and 3)
    default  the reference's default step: IMAGE_SIZE 512, 8 random crops of 256 x 256, through the product's host loop
    multilevel  config 2's "16-level grid" as the extension it is (multilevel.py: the reference reads ONE level pair per sample): 3840 x 2160,
             L = 5 level pairs (10 grids, what the reference's halving pyramid holds at this size) concatenated into Cin = 361, "4 x 64" decoder;
             layer-wise kernels (nic_encode, nic_decoder_general_*, nic_encode_backward, nic_adam_multi), the image walked in chunks
    fits8    config 5, one GPU's share: 8 independent 1080p fits, each with its own grids / decoder / optimiser state and stream -
             concurrently with an eighth of the CUs each (nic_path_desc.max_workgroups) against the same 8 steps back to back
"""
import json
import math
import os
import time

import numpy as np
import torch

PEAK_HBM_GBS = 8000.0
PEAK_FP32_MATRIX_TFLOPS = 157.3


def _work(dim, method):
    k0 = 4 if (dim == 2 or method == 4) else 8
    k1 = 4 if dim == 2 else 8
    cin = 12 * (k0 + 1) + 6 * dim + 1
    flop = 6 * (cin * 64 + 64 * 64 + 3 * 64)
    byt = (k0 + k1) * 12 * 4 + 2 * (k0 + k1) * 12 * 4 + 3 * 4          # fp32 parameters / gradients / targets (SURVEY 8d formula)
    return cin, flop, byt


def _pct(xs):
    xs = np.asarray(xs, dtype=np.float64)
    return {"n": int(xs.size), "median": round(float(np.median(xs)), 4), "p10": round(float(np.percentile(xs, 10)), 4),
            "p90": round(float(np.percentile(xs, 90)), 4), "min": round(float(xs.min()), 4)}


def _fit_state(dev, dim, method, grid_base, seed=0, shapes=None):
    from neural_image_compression_v2_amd import fp_def
    from neural_image_compression_v2_amd.image_compression import ColorDecoder
    torch.manual_seed(seed)
    cin, _, _ = _work(dim, method)
    if shapes is not None:                                   # explicit node counts (sizes that are not multiples of 8 samples)
        fp = [torch.rand(*sh, device=dev) - 0.498 for sh in shapes]
    else:
        mk = fp_def.create_pyramid if dim == 2 else fp_def.create_pyramid_3d
        fp, _ = mk(grid_base, 12, 8, dev, torch.float32, True)
    dec = ColorDecoder(cin, 64).to(dev)
    params = [p.detach() for p in dec.linear_params()]
    tensors = params + [fp[0].detach(), fp[1].detach()]
    return {"g0": fp[0].detach(), "g1": fp[1].detach(), "params": params, "tensors": tensors,
            "m": [torch.zeros_like(t) for t in tensors], "v": [torch.zeros_like(t) for t in tensors], "flat": None}


def _adam(lib, _lib, st, out, i, total, stream):
    lrs = [0.005] * 6 + [0.01, 0.01]
    q_lo = -(2 ** 8 - 1) / 2 ** 9
    cos = 0.5 * (1 + math.cos(math.pi * i / max(total, 1)))
    grads = out.grad_mlp + [out.grad_g0, out.grad_g1]
    tab = (_lib.NicAdamTensor * 8)()
    for k, (p, g, m, v) in enumerate(zip(st["tensors"], grads, st["m"], st["v"])):
        lo, hi = (q_lo, 0.5) if k >= 6 else (1.0, -1.0)
        tab[k] = _lib.NicAdamTensor(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), i + 1, lrs[k] * cos, lo, hi)
    _lib.check(lib.nic_adam_multi(tab, 8, 0.9, 0.999, 1e-8, stream), "nic_adam_multi")


_DT = {"split": "bf16x2-split operands, f32 accumulate", "f32": "f32", "bf16": "bf16 operands, f32 accumulate", "fp16": "fp16 operands, f32 accumulate"}


def _run_volume(args, dev, name, dim, method, extent, grid_base, origin, split=True, shapes=None):
    """one crop = the whole volume (or slab) per step; bench.Fit holds the state (fp32 masters, optional 16-bit mirrors: --grid-dtype; --decoder; --precision)"""
    import bench
    from neural_image_compression_v2_amd import _lib
    gdt = {"f32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}[args.grid_dtype]
    fit = bench.Fit(dev, dim, method, grid_base=grid_base, shapes=shapes, n_linear=args.decoder, precision=args.precision, grid_dtype=gdt)
    n = int(np.prod(extent))
    g = torch.Generator(device="cpu").manual_seed(1234)
    target = torch.rand(n, 3, generator=g).to(dev)
    org = [list(origin)]
    total = args.warmup + args.steps

    def step(i, ev=None):
        geo = fit.geometry(i, extent, aligned=all(o % 8 == 0 for o in origin))
        out = fit.fwd_bwd(geo, org, target, ev, adam=(i, total))
        fit.adam(out, i, total)
        return out

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    ev = [bench.KernelEvents(fit.lib) for _ in range(args.steps)]
    t0 = time.perf_counter()
    trace = [t0]
    for i in range(args.steps):
        out = step(args.warmup + i, ev[i])
        trace.append(time.perf_counter())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    if os.environ.get("NIC_BENCH_TRACE") == "1":
        import sys
        print("host ms per step:", " ".join(f"{(b - a) * 1e3:.2f}" for a, b in zip(trace, trace[1:])), f"| total {(time.perf_counter() - t0) * 1e3:.1f}", file=sys.stderr)
    kms = [e.elapsed_ms() for e in ev]
    gb = 4 if fit.mirror is None else 2
    cin, flop, byt = bench.work_per_sample(dim, method, args.decoder, gb)
    km = float(np.median(kms))
    unit = "Mvoxels/s" if dim == 3 else "Mpixels/s"
    gbs = byt * n / (km * 1e-3) / 1e9
    tfl = flop * n / (km * 1e-3) / 1e12
    return {"metric": f"{unit[:-2]}/sec train-step (fwd+bwd + Adam + clamp), {name}", "value": round(n / dt / 1e6, 2), "unit": unit, "n_gpus": 1,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": ("bf16x2-split products (weight-gradient operands split on read from fp32 images), f32 accumulate"
                                           if (args.precision == "split" and dim == 3) else _DT[args.precision]),
            "data": "synthetic",
            "config": {"workload": name, "samples_per_step": n, "extent": list(extent), "grids": [list(g_.shape) for g_ in fit.master],
                       "grid_storage": args.grid_dtype + (" (fp32 masters + Adam state)" if fit.mirror is not None else ""), "decoder_linear_layers": args.decoder,
                       "method": method, "cin": cin, "final_loss": round(float(out.loss), 6)},
            "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
                         "traffic": None, "kernel_ms": round(km, 4), "stats": _pct(kms), "bytes_per_sample": byt, "flop_per_sample": flop,
                         "samples_per_launch": n, "kernel": bench.kernel_name(dim, method, args.precision, args.decoder),
                         **({"note": "above 1: SURVEY 8d's byte count assumes no reuse at all; neighbouring samples share their corners (64 samples per G0 cell), "
                                     "the grids sit in L2 / Infinity Cache and the kernel no longer pays those bytes - the figure stops being a bound here"} if gbs > PEAK_HBM_GBS else {}),
                         "mfma_f32_equivalent": {"achieved": round(tfl, 2), "peak": PEAK_FP32_MATRIX_TFLOPS, "unit": "TFLOP/s",
                                                 "frac": round(tfl / PEAK_FP32_MATRIX_TFLOPS, 4)}}}


def _time_loop(args, ic, graph):
    """seconds per step of the product's training loop: the host loop (train_step per step), or - graph > 0 - replays of a captured hipGraph of
    `graph` steps (train_models_graph: origins, noise offset, Adam step and learning rate from device memory)"""
    if graph:
        ic.train_models_graph(ic.feature_pyramid, steps_per_graph=graph, time_replays=True)
        t = ic.graph_timing
        return t["seconds"] / t["steps"], t["steps"], f"hipGraph replays of {graph} captured steps, no host work per step"
    for e in range(args.warmup):
        ic.train_step(ic.feature_pyramid, e)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for e in range(args.warmup, args.warmup + args.steps):
        ic.train_step(ic.feature_pyramid, e)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / args.steps, args.steps, "host loop included"


def _run_default(args, dev):
    """the reference's default step shape through the product's host loop (origins from the reference's RNG calls)"""
    import random
    from neural_image_compression_v2_amd.image_compression import ImageCompression
    from neural_image_compression_v2_amd.var2 import Settings
    graph = int(getattr(args, "graph", 0) or 0)
    n_ep = args.warmup + args.steps + 1 if not graph else int(math.ceil((1 + graph * (1 + (args.warmup + args.steps) // graph)) / 0.95)) + 1
    cfg = Settings(IMAGE_SIZE=512, NUM_EPOCHS=n_ep, TF_NO_MIP=True, TF_SPLIT_BF16=args.precision != "f32", TF_PLAIN_BF16=args.precision in ("bf16", "fp16"), TF_PLAIN_FP16=args.precision == "fp16",
                   TF_DEVICE_SAMPLER=bool(graph))
    S = cfg.IMAGE_SIZE
    u = torch.linspace(0, 1, S)
    img = torch.stack([0.5 + 0.25 * torch.sin(2 * math.pi * (c + 1) * u)[:, None] * torch.cos(2 * math.pi * (c + 2) * u)[None, :] for c in range(3)])
    ic = ImageCompression(cfg, dev, seed=0)
    ic.set_images([torch.round(img.clamp(0, 1) * 255).to(torch.uint8)])
    torch.manual_seed(1)
    random.seed(1)
    dt, steps, how = _time_loop(args, ic, graph)
    px = cfg.NUM_CROPS * 256 * 256
    cin, flop, byt = _work(2, 1)
    return {"metric": f"Mpixels/sec train-step, the reference's default 8 x 256^2 random-crop step ({how})", "value": round(px / dt / 1e6, 2),
            "unit": "Mpixels/s", "n_gpus": 1, "steps": steps, "warmup": args.warmup, "ms_per_step": round(dt * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": _DT[args.precision], "data": "synthetic",
            "config": {"workload": "IMAGE_SIZE 512, 8 random crops of 256 x 256 per step (var2.py defaults), targets from the resident uint8 image, "
                                   "one-launch Adam", "samples_per_step": px, "psnr_db": round(float(ic.psnr(ic.feature_pyramid)), 3)},
            "roofline": {"bound": "hbm", "achieved": round(byt * px / dt / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": round(byt * px / dt / 1e9 / PEAK_HBM_GBS, 4), "traffic": None, "note": "whole step (host loop) over the algorithmic bytes"}}


def _run_default3d(args, dev, method, size):
    """the shape of the reference's own 3D sweeps (the .bat launchers: COMPRESSION_METHOD 3 / 4, IMAGE_SIZE 64 / 128, CROP_MIP_LEVEL 5):
    8 random 32^3 crops of a resident uint8 volume per step, through the product's host loop"""
    import random
    from neural_image_compression_v2_amd.image_compression import ImageCompression
    from neural_image_compression_v2_amd.var2 import Settings
    graph = int(getattr(args, "graph", 0) or 0)
    cfg = Settings(IMAGE_SIZE=size, IMAGE_3D_SIZE=size, IMAGE_DIMENSION=3, COMPRESSION_METHOD=method, CROP_MIP_LEVEL=5,
                   NUM_EPOCHS=(args.warmup + args.steps + 1 if not graph else int(math.ceil((1 + graph * (1 + (args.warmup + args.steps) // graph)) / 0.95)) + 1),
                   TF_NO_MIP=True, TF_SPLIT_BF16=args.precision != "f32", TF_PLAIN_BF16=args.precision in ("bf16", "fp16"), TF_PLAIN_FP16=args.precision == "fp16", TF_DEVICE_SAMPLER=bool(graph))
    g = torch.Generator(device="cpu").manual_seed(5)
    vol = torch.randint(0, 256, (3, size, size, size), generator=g, dtype=torch.uint8)
    ic = ImageCompression(cfg, dev, seed=0)
    ic.set_images([vol], den=256.0)
    torch.manual_seed(1)
    random.seed(1)
    dt, steps, how = _time_loop(args, ic, graph)
    n = cfg.NUM_CROPS * 32 ** 3
    cin, flop, byt = _work(3, method)
    return {"metric": f"Mvoxels/sec train-step, the reference's 3D sweep shape: 8 x 32^3 random crops of a {size}^3 volume, method {method} ({how})",
            "value": round(n / dt / 1e6, 2), "unit": "Mvoxels/s", "n_gpus": 1, "steps": steps, "warmup": args.warmup, "ms_per_step": round(dt * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("bf16x2-split products (weight-gradient operands split on read from fp32 images), f32 accumulate" if args.precision == "split" else _DT[args.precision]),
            "data": "synthetic",
            "config": {"workload": f"IMAGE_SIZE {size}, IMAGE_DIMENSION 3, COMPRESSION_METHOD {method}, CROP_MIP_LEVEL 5, NUM_CROPS 8, targets from the resident "
                                   "uint8 volume, one-launch Adam", "samples_per_step": n, "cin": cin},
            "roofline": {"bound": "hbm", "achieved": round(byt * n / dt / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": round(byt * n / dt / 1e9 / PEAK_HBM_GBS, 4), "traffic": None, "note": "whole step (host loop) over the algorithmic bytes"}}


def _run_fits8(args, dev):
    """config 5 on one GPU: 8 independent 1080p fits; concurrent on 8 streams (an eighth of the CUs each) vs back to back (whole chip each)"""
    from neural_image_compression_v2_amd import _lib, fused
    lib = _lib.load()
    HH, WW, NF = 1080, 1920, 8
    fits, targets, streams = [], [], []
    g = torch.Generator(device="cpu").manual_seed(99)
    for k in range(NF):
        fits.append(_fit_state(dev, 2, 1, (HH // 4, WW // 4), seed=k))
        targets.append(torch.rand(HH * WW, 3, generator=g).to(dev))
        streams.append(torch.cuda.Stream(dev))
    org = [[0, 0]]
    total = args.warmup + args.steps
    cu = torch.cuda.get_device_properties(dev).multi_processor_count

    def one(k, i, max_wg):
        st = fits[k]
        geo = fused.PathGeometry(dim=2, method=1, step_number=0.25, mip_level=0, extent=(HH, WW), num_crops=1, noise_mode=_lib.NIC_NOISE_KERNEL,
                                 noise_seed=7 + k, noise_offset=i, split_bf16=args.precision == "split", bf16=args.precision == "bf16", fp16=args.precision == "fp16", max_workgroups=max_wg)
        out = fused.fused_forward_backward(geo, st["g0"], st["g1"], org, st["params"], targets[k], flat=st["flat"])
        st["flat"] = out.flat
        _adam(lib, _lib, st, out, i, total, _lib.stream_ptr(dev))
        return out

    res = {}
    # concurrent_N: 8 streams, every launch limited to cu / N workgroups (N fits side by side fill the chip; ROCm maps streams onto
    # GPU_MAX_HW_QUEUES hardware queues, 4 by default, so N = 8 needs that raised to run all eight at once)
    for mode in ("back_to_back", "concurrent_2", "concurrent_4", "concurrent_8"):
        share = 0 if mode == "back_to_back" else cu // int(mode.split("_")[1])

        def sweep(i):
            outs = []
            for k in range(NF):
                if share:
                    with torch.cuda.stream(streams[k]):
                        outs.append(one(k, i, share))
                else:
                    outs.append(one(k, i, 0))
            return outs
        for i in range(args.warmup):
            sweep(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            outs = sweep(args.warmup + i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        res[mode] = {"ms_per_sweep": round(dt * 1e3, 4), "mpix_s": round(NF * HH * WW / dt / 1e6, 2), "loss_fit0": round(float(outs[0].loss), 6)}
    best = max(res, key=lambda m: res[m]["mpix_s"])
    cin, flop, byt = _work(2, 1)
    return {"metric": "aggregate Mpixels/sec, 8 independent 1080p fits on one MI355X (BASELINE config 5, one GPU's share)", "value": res[best]["mpix_s"],
            "unit": "Mpixels/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": res[best]["ms_per_sweep"], "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16x2-split operands, f32 accumulate", "data": "synthetic",
            "config": {"workload": "8 x (1920 x 1080 fit, own grids [12,481,271] + [12,241,136], own decoder, own Adam state), one step of each per sweep",
                       "modes": res, "reported": best, "GPU_MAX_HW_QUEUES": __import__("os").environ.get("GPU_MAX_HW_QUEUES", "default (4)")},
            "roofline": {"bound": "hbm", "achieved": round(byt * res[best]["mpix_s"] * 1e6 / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": round(byt * res[best]["mpix_s"] * 1e6 / 1e9 / PEAK_HBM_GBS, 4), "traffic": None}}


def _run_multilevel(args, dev, levels=5, channels=4, n_linear=3, layerwise=False, chunk=(1024, 540)):
    """one step = one pass over every pixel of a 3840 x 2160 image + one Adam launch over all 2 L grids and the decoder.  Fused route (the default where a
    kernel exists, fused.ml_is_fused): ONE launch for the whole pass - gathers of every pair, in-kernel noise, plain-bf16 decoder, loss, backward, one gradient
    flush per touched cell and pair (csrc/fused_q16.hpp::QML).  ``layerwise``: round 3's composition of nic_encode x L + the general decoder, in chunks."""
    from neural_image_compression_v2_amd.multilevel import MultiLevelField
    H, W = 2160, 3840
    f = MultiLevelField((H, W), levels, channels=channels, hidden=64, n_linear=n_linear, device=dev, seed=0, fused_step=False if layerwise else None)
    tgt = torch.rand(H, W, 3, device=dev)
    chunks = [(x0, y0) for x0 in range(0, H, chunk[1]) for y0 in range(0, W, chunk[0])]
    flat_t = tgt.reshape(-1, 3)

    def step():
        if f.fused_step:
            return f.train_step([[0, 0]], (H, W), flat_t, noise=True)
        tot = None
        for k, (x0, y0) in enumerate(chunks):
            ext = (min(chunk[1], H - x0), min(chunk[0], W - y0))
            t = tgt[x0:x0 + ext[0], y0:y0 + ext[1]].reshape(-1, 3)
            l = f.train_step([[x0, y0]], ext, t, noise=True, accumulate=k > 0, scale=ext[0] * ext[1] / (H * W), step=k == len(chunks) - 1)
            tot = l if tot is None else tot + l
        return tot

    for _ in range(max(args.warmup, 1)):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    px = H * W
    C, P = channels, 6
    byt = levels * (8 * C * 4 + 2 * 8 * C * 4) + 3 * 4                      # SURVEY 8d's extended-mode formula: L pairs x (K0 + K1) C x (e_param + 2 e_grad) + target
    flop = 6 * (f.cin * 64 + (n_linear - 2) * 64 * 64 + 3 * 64)
    route = "fused (one launch per step: nic_fused_ml_forward_backward, plain-bf16 products, in-kernel noise)" if f.fused_step else \
            f"layer-wise (nic_encode x L + general decoder, torch.rand noise, {len(chunks)} chunks of {chunk[1]}x{chunk[0]} px)"
    rec = {"metric": "Mpixels/sec train-step, multi-level extension", "value": round(px / dt / 1e6, 2), "unit": "Mpixels/s", "n_gpus": 1,
           "steps": args.steps, "warmup": max(args.warmup, 1), "ms_per_step": round(dt * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "bf16 operands, f32 accumulate" if f.fused_step else "f32", "data": "synthetic",
           "config": {"workload": f"3840x2160 RGB fit, every pixel once per step, {levels} level pairs of {C} channels ({2 * levels} grids "
                                  f"{[list(g.shape) for g in f.fp]}) concatenated: Cin {f.cin}, {n_linear}xLinear(64) decoder, noise, MSE, Adam + clamp; route: {route}",
                      "samples_per_step": px, "final_loss": round(float(loss), 6), "parameters": int(sum(g.numel() for g in f.fp)), "fused": bool(f.fused_step)},
           "roofline": {"bound": "hbm", "achieved": round(byt * px / dt / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(byt * px / dt / 1e9 / PEAK_HBM_GBS, 4),
                        "traffic": None, "bytes_per_sample": byt, "flop_per_sample": flop,
                        "note": "algorithmic bytes of the extended mode (element-granular, no reuse); whole step by wall clock, optimiser included"}}
    if not f.fused_step:
        rec["roofline"]["note"] += f"; this route MATERIALISES the [N, Cin] input, its gradient and every activation (~ {4 * (3 * f.cin + 4 * (n_linear - 1) * 64)} B per sample of real traffic)"
    return rec


def run(args):
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    w = args.workload
    recs = []
    if w == "lut33":
        recs.append(_run_volume(args, dev, "33^3 colour LUT, one crop, method 3 (reference weights), grids 10^3 + 6^3", 3, 3, (33, 33, 33), None, (0, 0, 0),
                                shapes=[(12, 10, 10, 10), (12, 6, 6, 6)]))
    elif w in ("vol64", "vol128"):
        S = 64 if w == "vol64" else 128
        for method in (3, 4):
            recs.append(_run_volume(args, dev, f"{S}^3 volume, one crop, method {method}", 3, method, (S, S, S), S // 4, (0, 0, 0)))
    elif w == "slab":
        for method in (4, 3):
            recs.append(_run_volume(args, dev, f"1920x1080x64 video field, rank 3's slab z in [720, 960) of 8, method {method}", 3, method, (64, 1080, 240),
                                    (16, 270, 480), (0, 0, 720)))
    elif w == "default":
        recs.append(_run_default(args, dev))
    elif w == "default3d":
        for size in (64, 128):
            for method in (3, 4):
                recs.append(_run_default3d(args, dev, method, size))
    elif w == "fits8":
        recs.append(_run_fits8(args, dev))
    elif w == "multilevel":
        # the fused kernels: 5 pairs x 4 channels (Cin 161) with the reference's 3-Linear decoder - what fits the LDS at L = 5 -, 3 pairs
        # with the north star's 5-Linear decoder, 3 pairs x 12 channels; then round 3's layer-wise composition at 5 x 12 (Cin 361, no fused kernel) for reference
        recs.append(_run_multilevel(args, dev, 5, 4, 3))
        recs.append(_run_multilevel(args, dev, 3, 4, 5))
        recs.append(_run_multilevel(args, dev, 3, 12, 3))
        if getattr(args, "ml_layerwise", False):
            recs.append(_run_multilevel(args, dev, 5, 12, 5, layerwise=True))
    else:
        raise SystemExit(f"unknown workload {w}")
    for r in recs:
        print(json.dumps(r), flush=True)
