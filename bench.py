#!/usr/bin/env python3
"""Headline benchmark: Mpixels/s of one full training step (fused forward + backward, gradient exchange, Adam + clamp)
on BASELINE.json's configs[1]: a 3840 x 2160 RGB fit, dense G0/G1 grid pair (reference semantics, no-mip),
3 x Linear(64) decoder, in-kernel Threefry noise - every pixel of the image once per step.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched under torch.distributed.run, one rank per GPU)

N > 1 is weak scaling (8.29 Mpx per rank and step).  Default --shard stripes: rank r owns a stripe of the image's second axis (a
contiguous block of grid node rows) and takes its N passes per step from it - the same sample multiset as N replicas each covering
the image once, but the step exchanges only the loss, the decoder gradients and one boundary node row per neighbour pair (0.3 MB at
N = 8) instead of the dense grid gradients (31 MB, --shard replicated); the full grids are assembled once after the timed region.

Prints ONE JSON line on rank 0 (see the driver contract); adds `roofline` (dominant kernel = fused_kernel, timed with HIP
events on its launch stream inside the timed region) and, at N = 1, `cpu_baseline` (the CPU oracle timed on a bounded
strip of the same workload).  --precision split (default): the 2D training default, every matrix product as hi + lo bf16 pairs on
the bf16 matrix pipe with fp32 accumulation (gradients within 5e-6 of the fp32 kernel; parity-tested against the CPU oracle at the
fp32 kernel's tolerances); --precision f32: v_mfma_f32_32x32x2_f32 throughout.
"""
import argparse
import ctypes
import json
import math
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

H, W = 2160, 3840                       # first sample axis ("x", coord[0]) = image axis 0, like the reference's [3, S, S] tensors
CIN, HID = 73, 64
FLOP_PER_SAMPLE = 6 * (CIN * HID + HID * HID + 3 * HID)          # fwd + bwd MACs x 2 = 53,760
BYTES_PER_SAMPLE = 8 * 12 * 4 + 2 * 8 * 12 * 4 + 3 * 4            # SURVEY 8d, fp32 params / fp32 grads / fp32 target = 1,164
PEAK_FP32_MATRIX_TFLOPS = 157.3                                   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak
PEAK_BF16_TFLOPS = 2500.0                                        # dense bf16 MFMA peak (same guide)
PEAK_HBM_GBS = 8000.0


def synthetic_target(device):
    """rgb = 1/2 + 1/4 sin(2 pi f_c u) cos(2 pi g_c v) + 0.05 U(-1,1), quantised to 8 bit; [N, 3] in sample order"""
    g = torch.Generator().manual_seed(1234)
    u = torch.linspace(0, 1, H).view(H, 1)
    v = torch.linspace(0, 1, W).view(1, W)
    chans = []
    for c in range(3):
        f, gq = 3.0 + 2 * c, 5.0 + 3 * c
        chans.append(0.5 + 0.25 * torch.sin(2 * math.pi * f * u) * torch.cos(2 * math.pi * gq * v))
    img = torch.stack(chans) + 0.05 * (torch.rand(3, H, W, generator=g) * 2 - 1)
    img = torch.floor(img.clamp(0, 1) * 255 + 0.5) / 255
    return img.permute(1, 2, 0).reshape(-1, 3).contiguous().to(device), img


def cpu_baseline(img, steps=3):
    """the CPU oracle (certified against the reference by tests/test_oracle_golden.py) on a bounded strip of the workload:
    the reference's op sequence - gathers, blend, PE, cat, rand_like noise, 3 Linear + GELU, MSE, autograd backward."""
    from oracle import nic_oracle as O
    strip = 256
    g = torch.Generator().manual_seed(0)
    fp, _ = O.create_pyramid((H // 4, W // 4), 12, 8, dim=2, no_mip=True, generator=g)
    mlp = O.init_mlp(CIN, HID, generator=g)
    tgt = img[:, :, :strip].permute(1, 2, 0).reshape(-1, 3).contiguous()
    n = H * strip
    times = []
    for _ in range(steps):
        t0 = time.perf_counter()
        x_noise_shape = (n, CIN)
        noise = (torch.rand(*x_noise_shape) - 0.5) / 256
        O.forward_backward(fp[0], fp[1], mlp, [(0, 0)], (H, strip), 0.25, 0, tgt, noise, 6)
        times.append(time.perf_counter() - t0)
    t = float(np.median(times))
    return {"value": round(n / t / 1e6, 4), "unit": "Mpixels/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{H}x{strip} strip of the 4K workload ({n} px) per step, median of {steps} steps of fwd+bwd (eager torch on host cores)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", choices=["split", "f32"], default="split",
                    help="split: every matrix product of the step as hi + lo bf16 pairs on the bf16 matrix pipe, fp32 accumulate (gradients "
                         "within 5e-6 of the fp32 kernel; the product's default for 2D training); f32: v_mfma_f32_32x32x2_f32 throughout")
    ap.add_argument("--target", choices=["tensor", "image"], default="tensor",
                    help="tensor: resident fp32 [N,3] targets (the reference's crop stack, built once); image: targets read from the "
                         "resident uint8 image inside the step (a quarter of the bytes, ~3 %% more kernel time: three byte gathers)")
    ap.add_argument("--shard", choices=["stripes", "replicated"], default="stripes",
                    help="N > 1.  stripes: every rank owns a stripe of the image's second axis (a contiguous block of grid node rows) and "
                         "takes its N passes per step from it; the step exchanges the loss, the decoder gradients and one boundary node "
                         "row per neighbour pair (~0.3 MB) instead of the dense grid gradients.  replicated: every rank covers the whole "
                         "image once per step and the whole gradient bucket (31 MB) is all-reduced")
    ap.add_argument("--virtual-world", type=int, default=0,
                    help="diagnostic, one process: run rank 0's share of an N-rank stripe-sharded step without the collectives")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0)) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        backend = os.environ.get("NIC_DIST_BACKEND", "nccl")         # "gloo": rehearsal of the N > 1 path on a box with one GPU
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))

    from neural_image_compression_v2_amd import _lib, fp_def, fused
    from neural_image_compression_v2_amd.distributed import all_reduce_flat, assemble_stripes, plan_stripes, stripe_exchange
    from neural_image_compression_v2_amd.image_compression import ColorDecoder
    lib = _lib.load()

    torch.manual_seed(0)                                              # same init on every rank (replicated parameters)
    fp, _ = fp_def.create_pyramid((H // 4, W // 4), 12, 8, dev, torch.float32, True)      # [12, 961, 541], [12, 481, 271]
    dec = ColorDecoder(CIN, HID).to(dev)
    params = [p.detach() for p in dec.linear_params()]
    g0, g1 = fp[0].detach(), fp[1].detach()
    target, img = synthetic_target(dev)
    if args.target == "image":                                        # 8-bit codes of the same image, resident: u / 255 = img exactly
        target = fused.TargetImage(torch.round(img * 255).to(torch.uint8).to(dev), 255.0)
    n_local = H * W
    vworld = args.virtual_world if world == 1 and args.virtual_world > 1 else world
    n_global = n_local * vworld
    stripes = vworld > 1 and args.shard == "stripes"
    if stripes:
        # rank r: the stripe [start, start + size) of image axis 1, `world` passes over it per step (nic_path_desc.passes)
        plan = plan_stripes(W, 8, rank, vworld)                       # G1 cell = 8 pixels at step 1/4
        extent, ncrops, passes = (H, plan.size), 1, vworld            # one crop, `world` passes: a cell's gradients leave the CU once
        org = torch.tensor([[0, plan.start]], dtype=torch.int32, device=dev)
        if args.target == "tensor":
            target = img[:, :, plan.start:plan.start + plan.size].permute(1, 2, 0).reshape(-1, 3).repeat(vworld, 1).contiguous().to(dev)
    else:
        extent, ncrops, passes = (H, W), 1, 1
        org = torch.zeros(1, 2, dtype=torch.int32, device=dev)
    n_mine = extent[0] * extent[1] * ncrops * passes                  # = n_local unless the stripes are uneven (W / 8 px not a multiple of N)
    base_mine = H * plan.start * vworld if stripes else rank * n_local   # global id of this rank's first sample
    offs, sizes, total = fused.grad_bucket_layout(fused.PathGeometry(2, 1, 0.25, 0, extent, ncrops), g0, g1)
    flat = torch.zeros(total, dtype=torch.float32, device=dev)
    tensors = params + [g0, g1]                                       # Adam state per tensor, order of the bucket (after the loss)
    m_state = [torch.zeros_like(t) for t in tensors]
    v_state = [torch.zeros_like(t) for t in tensors]
    lrs = [0.005] * 6 + [0.01, 0.01]                                  # image_compression.py:361-364
    q_lo = -(2 ** 8 - 1) / 2 ** 9
    total_steps = args.warmup + args.steps
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    stream = _lib.stream_ptr(dev)
    adam_tab = (_lib.NicAdamTensor * len(tensors))()

    def step(i, events=None):
        geo = fused.PathGeometry(dim=2, method=1, step_number=0.25, mip_level=0, extent=extent, num_crops=ncrops, passes=passes,
                                 noise_mode=_lib.NIC_NOISE_KERNEL, noise_seed=7, noise_offset=i,
                                 sample_base=base_mine, loss_scale=1.0 / (3.0 * n_global),
                                 flags=_lib.NIC_FLAG_ORIGINS_ALIGNED,  # origins are multiples of the G1 cell
                                 split_bf16=args.precision == "split")
        out = fused.fused_forward_backward(geo, g0, g1, org, params, target, flat=flat, events=events)
        if world > 1 and stripes:
            stripe_exchange(plan, out.flat[:offs[7]], out.grad_g0, out.grad_g1)   # RCCL sum of [loss | decoder grads | boundary rows]
        elif stripes:                                                 # --virtual-world: the pack / unpack launches without the collective
            stripe_exchange(plan, out.flat[:offs[7]], out.grad_g0, out.grad_g1, reduce=lambda t, g: None)
        else:
            all_reduce_flat(out.flat)                                 # RCCL sum of [loss | decoder grads | grid grads]
        cos = 0.5 * (1 + math.cos(math.pi * i / max(total_steps, 1)))  # CosineAnnealingLR(T_max), eta_min = 0
        grads = out.grad_mlp + [out.grad_g0, out.grad_g1]
        for k, (p, g, m, v) in enumerate(zip(tensors, grads, m_state, v_state)):
            lo, hi = (q_lo, 0.5) if k >= 6 else (1.0, -1.0)           # fp_quantize_clamp on the grids only
            adam_tab[k] = _lib.NicAdamTensor(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), i + 1, lrs[k] * cos, lo, hi, 0)
        _lib.check(lib.nic_adam_multi(adam_tab, len(tensors), 0.9, 0.999, 1e-8, stream), "nic_adam_multi")   # Adam + clamp: one launch
        return out

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(args.warmup + i, ev[i])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    loss = float(out.loss)
    if world > 1 and stripes:
        assemble_stripes(plan, g0, g1)                                # once, outside the timed region: the full grids on every rank
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))       # fused kernel (+ its ~10 us partial reduction)
    # the metric's "+ PSNR" (outside the timed region): decode the whole image with the current parameters, PSNR with peak 2^8
    # against the synthetic target (utils.py:117-130) - after warmup + steps optimiser steps from a random initialisation
    from neural_image_compression_v2_amd import models, utils
    dgeo = fused.PathGeometry(dim=2, method=1, step_number=0.25, mip_level=0, extent=(H, W), num_crops=1,
                              flags=_lib.NIC_FLAG_ORIGINS_ALIGNED, split_bf16=args.precision == "split")
    rec = fused.fused_forward(dgeo, g0, g1, torch.zeros(1, 2, dtype=torch.int32, device=dev), params)
    ref_img = img.permute(1, 2, 0).reshape(-1, 3).contiguous().to(dev)
    psnr = float(utils.calculate_psnr(models.quantize_to_bit(rec, 8), models.quantize_to_bit(ref_img, 8)))

    if rank == 0:
        mpix = n_local * world * args.steps / elapsed / 1e6
        if stripes:
            xb = 4 * (offs[7] + (vworld - 1) * 12 * (g0.shape[2] + g1.shape[2]))
            par = (f"dp{vworld}, grids sharded in stripes of image axis 1 ({plan.size} px = {vworld} passes per rank and step); exchange per step = "
                   f"loss + decoder grads + {vworld - 1} boundary node rows of G0 and G1 = {xb} B all-reduced"
                   + (" (virtual: one process, no collectives)" if world == 1 else ""))
        else:
            par = f"dp{world} (sample-sharded, replicated parameters" + (f", {4 * total} B all-reduced per step)" if world > 1 else ")")
        flops = FLOP_PER_SAMPLE * n_mine / (kern_ms * 1e-3) / 1e12
        gbs = BYTES_PER_SAMPLE * n_mine / (kern_ms * 1e-3) / 1e9
        traffic = None
        tp = os.path.join(ROOT, "profiles", "traffic.json")            # HBM bytes per launch from the rocprofv3 --pmc passes, if collected
        if os.path.exists(tp):
            try:
                traffic = json.load(open(tp)).get("fused_kernel_bytes_per_launch")
            except Exception:
                traffic = None
        res = {
            "metric": "Mpixels/sec train-step (fwd+bwd) + PSNR, 4K RGB, 1/2/4/8 MI355X",
            "value": round(mpix, 2), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == "f32" else "bf16x2-split operands, f32 accumulate", "data": "synthetic",
            "config": {"workload": "3840x2160 RGB fit, every pixel once per step: dense G0 [12,961,541] + G1 [12,481,271] grid pair "
                                   "(reference semantics, no-mip), triangular PE, 3xLinear(64) GELU decoder, in-kernel Threefry-4x32-12 noise, MSE, "
                                   "fused fwd+bwd + grad all-reduce + Adam + clamp",
                       "pixels_per_step_per_gpu": n_local, "parallelism": par,
                       "final_loss": round(loss, 6), "psnr_db_after_these_steps": round(psnr, 3)},
        }
        common = {"traffic": traffic, "kernel_ms": round(kern_ms, 4), "flop_per_sample": FLOP_PER_SAMPLE, "bytes_per_sample": BYTES_PER_SAMPLE,
                  "samples_per_launch": n_mine}
        hbm = {"achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4)}
        if args.precision == "f32":
            # fp32: the matrix pipe is the tighter roofline (157.3 TFLOP/s / 53 760 = 2.9 Gpx/s vs HBM 8 TB/s / 1 164 B = 6.9 Gpx/s)
            res["roofline"] = {"bound": "mfma", "achieved": round(flops, 2), "peak": PEAK_FP32_MATRIX_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(flops / PEAK_FP32_MATRIX_TFLOPS, 4), **common,
                               "kernel": "fused_kernel<Layout<1>, SRC_ENCODE, MODE_TRAIN_MSE, float, PREC_F32>", "hbm_algorithmic": hbm}
        else:
            # split bf16: three bf16 MFMAs per product -> matrix ceiling 2 500 / (3 x 53 760) = 15.5 Gpx/s; the algorithmic-HBM ceiling
            # (6.9 Gpx/s) is the tighter one, so it is the reported bound; the matrix-pipe figures ride beside it
            res["roofline"] = {"bound": "hbm", **hbm, **common,
                               "kernel": "fused_kernel<Layout<1>, SRC_ENCODE, MODE_TRAIN_MSE, float, PREC_SPLIT>",
                               "mfma_bf16": {"achieved_executed": round(3 * flops, 1), "achieved_algorithmic": round(flops, 2),
                                             "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac_executed": round(3 * flops / PEAK_BF16_TFLOPS, 4)}}
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(img)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
