#!/usr/bin/env python3
"""Headline benchmark: Mpixels/s of one full training step (fused forward + backward, gradient exchange, Adam + clamp)
on BASELINE.json's configs[1]: a 3840 x 2160 RGB fit, dense G0/G1 grid pair (reference semantics, no-mip),
3 x Linear(64) decoder, in-kernel Threefry noise - every pixel of the image once per step.

    python bench.py --gpus N --steps K --warmup W

N > 1: when no rank environment is present this process starts `python -m torch.distributed.run --nproc-per-node N bench.py ...`
as a CHILD (before anything touches the GPU), relays its output and exits with its code; under torch.distributed.run (the driver's
own launch) it is one rank and insists on WORLD_SIZE == --gpus.

Multi-GPU shapes (one process per GPU, RCCL):
  --scaling weak (default)    8.29 Mpx per RANK and step.  --shard stripes (default): rank r owns a stripe of the image's second axis (a
                              contiguous block of grid node rows) and takes its N passes per step from it - the same sample multiset as N
                              replicas each covering the image once - exchanging the loss, the decoder gradients and one boundary node row
                              per neighbour pair (0.3 MB at N = 8); --shard replicated: every rank covers the whole image and the whole
                              gradient bucket (31 MB) is all-reduced.
  --scaling strong            ONE 8.29 Mpx pass per step split N ways (the north star's "the pixel batch shards across the 8 GPUs"):
                              stripes: rank r takes its stripe once (passes = 1), same small exchange; replicated: rank r takes its
                              stripe of SAMPLES against replicated grids and the whole bucket is all-reduced (the all-reduce-bound variant).
  Under stripes a rank runs Adam over its own node rows only (distributed.stripe_param_blocks: 1 / N of the optimiser work, moments
  allocated per stripe); the full grids are assembled once after the timed region.
  --workload video --gpus N   BASELINE config 4: the 1920 x 1080 x 64 field, z-stripes of the 3D grids, same two scalings.
  --workload fits64 --gpus N  BASELINE config 5: 64 independent 1080p fits, 64 / N per GPU back to back (replicas only, no collective).

Prints ONE JSON line on rank 0 (see the driver contract).  `value` comes from the K timed steps alone.  Beside it:
  roofline       dominant kernel (the fused training kernel of --precision), HIP events on its launch stream: `kernel_ms` = mean over
                 the K timed steps (what `achieved` uses); `stats` = median / p10 / p90 over a separate leg of >= 100 launches run
                 after the timed region (the K = 20 of the driver's default is 45 ms - too short for percentiles)
  roofline_*     the other arithmetic modes measured in the same run on the same inputs (kernel-only legs): f32 / split / bf16
  roofline_4x64  the north star's literal configuration as a WHOLE step: 5-Linear "4 x 64" decoder, bf16 grid storage with fp32 masters,
                 plain-bf16 products, fused step + reduce + one nic_adam_multi over 10 + 2 tensors (16-bit mirrors rewritten)
  cpu_baseline   N = 1: the CPU oracle on a bounded strip of the same workload at the fastest thread count found on the box
--precision split (default): every matrix product as hi + lo bf16 pairs on the bf16 matrix pipe with fp32 accumulation (gradients within
5e-6 of the fp32 kernel; the headline stays on it so that it remains comparable with the reference's fp32 arithmetic); f32:
v_mfma_f32_32x32x2_f32; bf16: plain bf16 products (NIC_FLAG_BF16; checked against the precision-emulating oracle, tests/test_gpu_bf16.py).
"""
import argparse
import ctypes
import json
import math
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

H, W = 2160, 3840                       # first sample axis ("x", coord[0]) = image axis 0, like the reference's [3, S, S] tensors
VID = (64, 1080, 1920)                  # config 4: sample axes (x <-> T, y <-> H, z <-> W); grids [12, 481, 271, 17] + [12, 241, 136, 9]
HID = 64
PEAK_FP32_MATRIX_TFLOPS = 157.3                                   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak
PEAK_BF16_TFLOPS = 2500.0                                        # dense bf16 MFMA peak (same guide)
PEAK_HBM_GBS = 8000.0
TRAFFIC_SOURCE = "profiles/traffic.json"


def work_per_sample(dim, method, n_linear=3, grid_bytes=4, target_bytes=4):
    """Cin, FLOPs (fwd + bwd MACs x 2) and algorithmic bytes per sample (element-granular, no reuse: gathers at the storage
    width, fp32 atomic read-modify-writes of the gradient tables, the target)"""
    k0 = 4 if (dim == 2 or method == 4) else 8
    k1 = 4 if dim == 2 else 8
    cin = 12 * (k0 + 1) + 6 * dim + 1
    flop = 6 * (cin * HID + (n_linear - 2) * HID * HID + 3 * HID)
    byt = (k0 + k1) * 12 * grid_bytes + 2 * (k0 + k1) * 12 * 4 + 3 * target_bytes
    return cin, flop, byt


def kernel_name(dim, method, precision, n_linear):
    if precision in ("bf16", "fp16"):
        return f"fused_q16_kernel<QL<{method if dim == 3 else 1}>, MODE_TRAIN, NL = {n_linear}> (8 waves x 16 samples, plain {precision} products)"
    if dim == 2 and n_linear == 5:
        return "fused_mlpn_kernel<Layout<1>, MODE_TRAIN, 5> (4 waves x 16 samples, split-bf16 products)"
    if dim == 2:
        return ("fused_train16_kernel<Layout<1>, MODE_TRAIN> (8 waves x 16 samples, split-bf16 products)" if precision == "split"
                else "fused_kernel<Layout<1>, SRC_ENCODE, MODE_TRAIN, float, PREC_F32>")
    return f"fused_kernel<Layout<{method}>, SRC_ENCODE, MODE_TRAIN, float, {'PREC_CHAIN' if precision == 'split' else 'PREC_F32'}>"


DTYPE_NAME = {"split": "bf16x2-split operands, f32 accumulate", "f32": "f32", "bf16": "bf16 operands, f32 accumulate", "fp16": "fp16 operands, f32 accumulate"}


def synthetic_image():
    """rgb = 1/2 + 1/4 sin(2 pi f_c u) cos(2 pi g_c v) + 0.05 U(-1,1), quantised to 8 bit; [3, H, W] on the host"""
    g = torch.Generator().manual_seed(1234)
    u = torch.linspace(0, 1, H).view(H, 1)
    v = torch.linspace(0, 1, W).view(1, W)
    chans = []
    for c in range(3):
        f, gq = 3.0 + 2 * c, 5.0 + 3 * c
        chans.append(0.5 + 0.25 * torch.sin(2 * math.pi * f * u) * torch.cos(2 * math.pi * gq * v))
    img = torch.stack(chans) + 0.05 * (torch.rand(3, H, W, generator=g) * 2 - 1)
    return torch.floor(img.clamp(0, 1) * 255 + 0.5) / 255


def synthetic_video(dev):
    """a smooth + detail field like the image's, [3, 64, 1080, 1920] uint8 codes, built on the device (132.7 Mvox)"""
    T, Hh, Ww = VID
    g = torch.Generator(device=dev).manual_seed(1234)
    t = torch.linspace(0, 1, T, device=dev).view(T, 1, 1)
    u = torch.linspace(0, 1, Hh, device=dev).view(1, Hh, 1)
    v = torch.linspace(0, 1, Ww, device=dev).view(1, 1, Ww)
    chans = []
    for c in range(3):
        f, gq = 3.0 + 2 * c, 5.0 + 3 * c
        ch = 0.5 + 0.25 * torch.sin(2 * math.pi * (f * u + 0.5 * t)) * torch.cos(2 * math.pi * gq * v) + 0.05 * (torch.rand(T, Hh, Ww, generator=g, device=dev) * 2 - 1)
        chans.append(torch.floor(ch.clamp(0, 1) * 255 + 0.5).to(torch.uint8))
    return torch.stack(chans)


def cpu_baseline(img, steps=5, budget_s=30.0):
    """the CPU oracle (certified against the reference by tests/test_oracle_golden.py) on a bounded strip of the workload:
    the reference's op sequence - gathers, blend, PE, cat, rand_like noise, 3 Linear + GELU, MSE, autograd backward.  Eager torch
    oversubscribes badly on many-core hosts (128 threads gave 0.19 Mpix/s where 8 cores of the survey box gave 0.4), so the
    thread count is probed first (one step each on a quarter strip) and the `steps` timed steps run at the fastest one."""
    from oracle import nic_oracle as O
    g = torch.Generator().manual_seed(0)
    fp, _ = O.create_pyramid((H // 4, W // 4), 12, 8, dim=2, no_mip=True, generator=g)
    mlp = O.init_mlp(73, HID, generator=g)

    def one(strip):
        tgt = img[:, :, :strip].permute(1, 2, 0).reshape(-1, 3).contiguous()
        n = H * strip
        t0 = time.perf_counter()
        noise = (torch.rand(n, 73) - 0.5) / 256
        O.forward_backward(fp[0], fp[1], mlp, [(0, 0)], (H, strip), 0.25, 0, tgt, noise, 6)
        return n, time.perf_counter() - t0

    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    t_start = time.perf_counter()
    cands = sorted({c for c in (4, 8, 16, 32, 64, 128) if 1 <= c <= avail} | ({avail} if avail < 4 else set()))
    probe = {}
    one(64)                                                            # page in, allocator warm-up
    for c in cands:
        torch.set_num_threads(c)
        n, dt = one(64)
        probe[c] = n / dt
        # eager torch collapses when oversubscribed (256 threads: 0.006 Mpix/s): stop once the rate has fallen well below the best
        if probe[c] < 0.7 * max(probe.values()) or time.perf_counter() - t_start > budget_s / 3:
            break
    best = max(probe, key=probe.get)
    torch.set_num_threads(best)
    strip, times = 256, []
    for _ in range(steps):
        n, dt = one(strip)
        times.append(dt)
        if time.perf_counter() - t_start > budget_s and len(times) >= 5:
            break
    t = float(np.median(times))
    return {"value": round(n / t / 1e6, 4), "unit": "Mpixels/s", "cores": best, "kind": "port",
            "threads_probed_mpix_s": {str(k): round(v / 1e6, 4) for k, v in probe.items()}, "host_cores_available": avail,
            "p10_p90_mpix_s": [round(n / float(np.percentile(times, 90)) / 1e6, 4), round(n / float(np.percentile(times, 10)) / 1e6, 4)],
            "sample": f"{H}x{strip} strip of the 4K workload ({n} px) per step, median of {len(times)} steps of fwd+bwd "
                      f"(eager torch, {best} threads = the fastest of {sorted(probe)} probed on this host)"}


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n):
    """--gpus N without a rank environment: become the launcher.  Nothing in this process has touched the GPU (no torch.cuda call
    so far), the ranks are CHILD processes, and their exit code is ours."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")                # dmabuf IPC for RCCL on this pool
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def pct(xs):
    xs = np.asarray(xs, dtype=np.float64)
    return {"n": int(xs.size), "median": round(float(np.median(xs)), 4), "p10": round(float(np.percentile(xs, 10)), 4),
            "p90": round(float(np.percentile(xs, 90)), 4), "min": round(float(xs.min()), 4), "mean": round(float(xs.mean()), 4)}


ISSUE_SOURCE = "profiles/issue.json"


def issue_record(kernel):
    """What actually binds these kernels: per-SIMD busy fractions from the committed rocprofv3 --pmc passes of the SAME kernel
    (profiles/issue.json, written by ab/collect_profiles.py; not measured in this run): vector ALU (4 cycles per wave instruction), matrix pipe,
    the share of wave-cycles spent waiting, the share of LDS cycles lost to bank conflicts."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "issue.json"))).get(kernel)
    except Exception:
        rec = None
    if rec is None:
        return None
    return {**rec, "issue_source": ISSUE_SOURCE + " (rocprofv3 --pmc passes of this kernel, committed with the tree; not measured in this run)"}


def cell_granular_bytes(dim, method, n_launch, grid_nodes, param_bytes=4, target_bytes=12, channels=12):
    """HBM bytes of one launch if every touched grid node is read once and its gradient read-modify-written once (what the cell-major kernels do:
    a G0 cell's 16 / 64 samples share their corners) + the targets: the bound the element-granular figure of SURVEY 8d overstates by the reuse"""
    nodes = int(sum(grid_nodes))
    return channels * nodes * (param_bytes + 8) + n_launch * target_bytes


def roofline_record(dim, method, precision, n_linear, kern_ms, n_launch, traffic=None, stats=None, grid_bytes=4, grid_nodes=None):
    """SURVEY 8d figures x the samples one launch processes / the kernel's launch duration.  The binding roofline: fp32 products -> the
    fp32 matrix pipe (157.3 TFLOP/s / 53 760 = 2.9 Gpx/s against HBM 8 TB/s / 1 164 B = 6.9 Gpx/s); bf16-pipe products (split: three
    MFMAs per product; plain: one) -> the algorithmic-HBM figure.  `frac_bf16_storage` is the same speed priced at the bytes
    of 16-bit grid storage (966 B per sample in 2D) - what the fraction would be on the north star's bf16 parameters."""
    cin, fl, byt = work_per_sample(dim, method, n_linear, grid_bytes)
    _, _, byt16 = work_per_sample(dim, method, n_linear, 2, 2)
    flops = fl * n_launch / (kern_ms * 1e-3) / 1e12
    gbs = byt * n_launch / (kern_ms * 1e-3) / 1e9
    common = {"traffic": traffic, "kernel_ms": round(kern_ms, 4), "flop_per_sample": fl, "bytes_per_sample": byt,
              "samples_per_launch": n_launch, "kernel": kernel_name(dim, method, precision, n_linear)}
    if traffic is not None:
        common["traffic_source"] = TRAFFIC_SOURCE + " (rocprofv3 --pmc passes of this tree's split kernel, profiles/r04_z_pmc.csv; not measured in this run)"
    if stats is not None:
        common["stats"] = stats
    iss = issue_record(common["kernel"])
    if iss is not None:
        common["issue"] = iss
        if traffic is None and iss.get("hbm_bytes_per_launch") is not None:      # this kernel's own --pmc passes (same launch shape: the 4K workload)
            common["traffic"] = int(iss["hbm_bytes_per_launch"])
            common["traffic_source"] = ISSUE_SOURCE + f" ({iss.get('pmc')}; not measured in this run)"
    if grid_nodes is not None:
        cb = cell_granular_bytes(dim, method, n_launch, grid_nodes, grid_bytes)
        cg = cb / (kern_ms * 1e-3) / 1e9
        common["hbm_cell_granular"] = {"bytes_per_launch": cb, "achieved": round(cg, 1), "unit": "GB/s", "frac": round(cg / PEAK_HBM_GBS, 4),
                                       "note": "every touched node read once + its gradient read-modify-written once + targets: the traffic a cell-major kernel "
                                               "needs; `frac` (element-granular, SURVEY 8d) prices every sample's 8 corner reads and 16 gradient updates separately"}
    hbm = {"achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
           "frac_bf16_storage": round(byt16 * n_launch / (kern_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4), "bytes_per_sample_bf16_storage": byt16}
    if precision == "f32":
        return {"bound": "mfma", "achieved": round(flops, 2), "peak": PEAK_FP32_MATRIX_TFLOPS, "unit": "TFLOP/s",
                "frac": round(flops / PEAK_FP32_MATRIX_TFLOPS, 4), **common, "hbm_algorithmic": hbm}
    mult = 3 if precision == "split" else 1
    return {"bound": "hbm", **hbm, **common,
            "note": "algorithmic bytes: the grids live in L2 / Infinity Cache (measured HBM traffic is ~4 % of them), the kernel is issue-bound",
            "mfma_bf16": {"achieved_executed": round(mult * flops, 1), "achieved_algorithmic": round(flops, 2), "peak": PEAK_BF16_TFLOPS,
                          "unit": "TFLOP/s", "frac_executed": round(mult * flops / PEAK_BF16_TFLOPS, 4)}}


class KernelEvents:
    """(start, end) pair of raw HIP events around the FUSED KERNEL ALONE inside a whole training step: `start` is recorded on the launch stream
    right before the entry point, `end` by the library right after its fused kernel - before the reduction / optimiser-tail launch
    (nic_mark_kernel_end).  Drop-in for the (torch.cuda.Event, torch.cuda.Event) pairs ``fused_forward_backward(events=)`` records."""
    _hip = None
    _inflight = []                 # pairs recorded and not yet waited for, oldest first
    MAX_INFLIGHT = 16              # the host stays at most this many timed steps ahead of the GPU (a bounded number of recorded-but-unfinished events)

    def __init__(self, lib):
        if KernelEvents._hip is None:
            KernelEvents._hip = ctypes.CDLL("libamdhip64.so")             # the runtime torch has already loaded
        hip = KernelEvents._hip
        self.lib = lib
        self.e = [ctypes.c_void_p(), ctypes.c_void_p()]
        for e in self.e:
            if hip.hipEventCreate(ctypes.byref(e)) != 0:
                raise RuntimeError("hipEventCreate failed")

    class _Rec:
        def __init__(self, owner, first):
            self.owner, self.first = owner, first

        def record(self, stream):
            if self.first:
                o = self.owner
                q = KernelEvents._inflight
                if len(q) >= KernelEvents.MAX_INFLIGHT:
                    KernelEvents._hip.hipEventSynchronize(q.pop(0).e[1])
                q.append(o)
                if KernelEvents._hip.hipEventRecord(o.e[0], ctypes.c_void_p(stream.cuda_stream)) != 0:
                    raise RuntimeError("hipEventRecord failed")
                o.lib.nic_mark_kernel_end(o.e[1])

    def __getitem__(self, i):
        return KernelEvents._Rec(self, i == 0)

    def elapsed_ms(self) -> float:
        if self in KernelEvents._inflight:
            KernelEvents._inflight.remove(self)
        ms = ctypes.c_float()
        if KernelEvents._hip.hipEventElapsedTime(ctypes.byref(ms), self.e[0], self.e[1]) != 0:
            raise RuntimeError("hipEventElapsedTime failed (an event was never recorded)")
        return float(ms.value)


class Fit:
    """One coordinate-network fit on one GPU: grids (fp32 masters, optional 16-bit mirrors the kernels gather from), decoder, Adam state,
    the reused gradient bucket and the one-launch optimiser table (built once, step counts and learning rates rewritten per step)."""

    def __init__(self, dev, dim, method, grid_base=None, shapes=None, n_linear=3, precision="split", grid_dtype=torch.float32, seed=0):
        from neural_image_compression_v2_amd import _lib, fp_def, fused
        from neural_image_compression_v2_amd.image_compression import ColorDecoder
        self._lib, self.fused, self.lib = _lib, fused, _lib.load()
        self.dev, self.dim, self.method, self.nl, self.precision = dev, dim, method, n_linear, precision
        self.cin = work_per_sample(dim, method)[0]
        torch.manual_seed(seed)                                       # same init on every rank (replicated parameters)
        if shapes is not None:
            q_lo = -(2 ** 8 - 1) / 2 ** 9
            fp = [(0.5 - q_lo) * torch.rand(*sh, device=dev) + q_lo for sh in shapes]
        else:
            mk = fp_def.create_pyramid if dim == 2 else fp_def.create_pyramid_3d
            fp, _ = mk(grid_base, 12, 8, dev, torch.float32, True)
        self.master = [fp[0].detach(), fp[1].detach()]
        self.mirror = None if grid_dtype == torch.float32 else [m.to(grid_dtype) for m in self.master]
        dec = ColorDecoder(self.cin, HID, n_linear).to(dev)
        self.params = [p.detach() for p in dec.linear_params()]
        self.flat = None
        self.table = None
        self._clean = False                                           # True: the grid-gradient part of the bucket is known to be zero
        self._tail_done = False
        self._tail_obj = None
        self._plans, self._orgs = {}, {}
        self.plan = None                                              # StripePlan: Adam over this rank's node rows only

    @property
    def grids(self):
        return self.mirror if self.mirror is not None else self.master

    def geometry(self, i, extent, ncrops=1, passes=1, sample_base=0, n_global=None, aligned=True, precision=None):
        _lib = self._lib
        pr = precision or self.precision
        return self.fused.PathGeometry(dim=self.dim, method=self.method, step_number=0.25, mip_level=0, extent=tuple(extent), num_crops=ncrops,
                                       passes=passes, noise_mode=_lib.NIC_NOISE_KERNEL, noise_seed=7, noise_offset=i, sample_base=sample_base,
                                       loss_scale=None if n_global is None else 1.0 / (3.0 * n_global),
                                       flags=_lib.NIC_FLAG_ORIGINS_ALIGNED if aligned else 0, split_bf16=pr == "split", bf16=pr == "bf16", fp16=pr == "fp16")

    def fwd_bwd(self, geo, org, target, events=None, adam=None):
        """``adam`` = (i, total_steps): the optimiser step of this training step rides on the reduction launch (nic_path_desc.tail) - two launches
        per step; the caller's ``adam(out, i, total_steps)`` that follows then launches nothing.  (Stripe-sharded fits exchange gradients between
        the two: their optimiser stays a launch of its own.)"""
        g0, g1 = self.grids
        use_tail = adam is not None and self.plan is None and os.environ.get("NIC_NO_TAIL") != "1"
        if geo.noise_mode != self._lib.NIC_NOISE_TENSOR and os.environ.get("NIC_NO_PLAN") != "1":
            # the steady state of a loop: one prepared launch per (shape, arithmetic, target) - fused.StepPlan, what the host loop of
            # ImageCompression.train_models uses: descriptor, parameter structs, gradient bucket and workspace are built once, a step rewrites
            # the noise offset and launches (the general wrapper re-validates everything per call: ~ 60 us of Python, more than a small step's GPU time)
            if not isinstance(org, torch.Tensor):
                okey = tuple(tuple(int(v) for v in o) for o in org)
                if okey not in self._orgs:
                    host = torch.tensor(okey, dtype=torch.int64).reshape(-1, geo.dim)
                    self.fused.check_origins(geo, host, g0, g1)
                    self._orgs[okey] = host.to(torch.int32).to(self.dev)
                org = self._orgs[okey]
            key = (tuple(geo.extent), geo.num_crops, geo.passes, geo.sample_base, geo.loss_scale, geo.flags, geo.split_bf16, geo.bf16, geo.fp16,
                   geo.dz_scale_log2, id(target), g0.data_ptr())
            plan = self._plans.get(key)
            if plan is None:
                plan = self._plans[key] = self.fused.StepPlan(geo, g0, g1, self.params, target)
            tail = self._tail(plan.gg0, plan.gg1, plan.gm, *adam) if use_tail else None
            plan.clean = self._clean and self.flat is plan.flat       # the optimiser zeroed what it read of THIS bucket (NIC_ADAM_ZERO_GRAD)
            out = plan.run(org, geo.noise_mode, geo.noise_seed, geo.noise_offset, tail=tail, events=events)
        else:
            tail = (lambda gg0, gg1, gm: self._tail(gg0, gg1, gm, *adam)) if use_tail else None
            out = self.fused.fused_forward_backward(geo, g0, g1, org, self.params, target, flat=self.flat, events=events, tail=tail, clean=self._clean)
        self.flat = out.flat
        self._clean = False
        if self._tail_done:
            self._clean = True                                        # the tail zeroed the grid gradients it read (NIC_ADAM_ZERO_GRAD)
        return out

    class _Tail:
        def __init__(self, fit, arr, n_stream):
            self.fit, self.arr = fit, arr
            self.struct = fit._lib.NicStepTail()
            self.struct.tensors = ctypes.cast(arr, ctypes.c_void_p).value
            self.struct.count, self.struct.n_stream = len(arr), n_stream
            self.struct.beta1, self.struct.beta2, self.struct.eps = 0.9, 0.999, 1e-8
            self.struct_ptr = ctypes.addressof(self.struct)

        def commit(self):
            self.fit._tail_done = True

    def _tail(self, gg0, gg1, gm, i, total_steps):
        if self.table is None or self._bucket != gg0.data_ptr():
            self._build_table(gm, gg0, gg1)
        if len(self.table) != 1:
            return None
        arr, lrs = self.table[0]
        if self._tail_obj is None or self._tail_obj.arr is not arr:
            self._tail_obj = Fit._Tail(self, arr, 2)                  # entries 0, 1: the grids (streamed); the rest: the decoder
        cos = 0.5 * (1 + math.cos(math.pi * i / max(total_steps, 1)))
        for k, lr in enumerate(lrs):
            arr[k].step = i + 1
            arr[k].lr = lr * cos
        return self._tail_obj

    def _build_table(self, gm, gg0, gg1):
        """nic_adam_tensor tables.  One fit on one GPU: [G0, G1 (whole, zeroing their gradient buckets as they go) | the decoder tensors] - one
        launch, or the tail of the step's reduction.  Stripe-sharded: TWO tables - the rank's INTERIOR node rows of both grids (one entry per grid:
        `reps` = channels runs of rows, moments allocated for the own rows only), whose gradients are final when the fused kernel ends and which
        therefore run under the all-reduce (``adam_interior``), and the boundary rows + the decoder, which need its result (``adam``)."""
        _lib = self._lib
        q_lo = -(2 ** 8 - 1) / 2 ** 9
        keep = []

        def entry(p_ptr, g_ptr, m_ptr, v_ptr, n, lr, clamp, mir, off16=0, zero=False, reps=0, stride=0, sstride=0):
            return _lib.NicAdamTensor(p_ptr, g_ptr, m_ptr, v_ptr, n, 0, lr, clamp[0], clamp[1], 0 if mir is None else mir.data_ptr() + 2 * off16,
                                      0 if mir is None else (1 if mir.dtype == torch.bfloat16 else 2), _lib.NIC_ADAM_ZERO_GRAD if zero else 0,
                                      reps, 0, stride, sstride)
        dec = []
        for p, g in zip(self.params, gm):
            m, v = torch.zeros_like(p), torch.zeros_like(p)
            keep += [m, v]
            assert p.is_contiguous() and g.is_contiguous()
            dec.append((entry(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), 0.005, (1.0, -1.0), None), 0.005))
        grids, interior, boundary = [], [], []
        for level, (p, g) in enumerate(zip(self.master, (gg0, gg1))):
            mir = None if self.mirror is None else self.mirror[level]
            assert p.is_contiguous() and g.is_contiguous() and g.shape == p.shape and (mir is None or (mir.is_contiguous() and mir.shape == p.shape))
            if self.plan is None:
                m, v = torch.zeros_like(p), torch.zeros_like(p)
                keep += [m, v]
                grids.append((entry(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), 0.01, (q_lo, 0.5), mir, zero=True), 0.01))
                continue
            from neural_image_compression_v2_amd.distributed import stripe_row_parts, stripe_state
            m, v = stripe_state(self.plan, level, p), stripe_state(self.plan, level, p)      # [C, own rows, ..]
            keep += [m, v]
            C, plane, row = int(p.shape[0]), int(p[0].numel()), int(p[0, 0].numel())
            lo_own = self.plan.node_rows(level)[0]
            splane = int(m[0].numel())

            def rows(r0, r1):       # rows r0 .. r1 (inclusive) of every channel: one entry
                o, so, n = r0 * row, (r0 - lo_own) * row, (r1 - r0 + 1) * row
                return (entry(p.data_ptr() + 4 * o, g.data_ptr() + 4 * o, m.data_ptr() + 4 * so, v.data_ptr() + 4 * so, n, 0.01, (q_lo, 0.5), mir, off16=o,
                              zero=True, reps=C, stride=plane, sstride=splane), 0.01)
            (ilo, ihi), brows = stripe_row_parts(self.plan, level)
            if ilo <= ihi:
                interior.append(rows(ilo, ihi))
            boundary += [rows(r, r) for r in brows]

        def table(ents):
            assert len(ents) <= _lib.NIC_ADAM_MAX_TENSORS
            arr = (_lib.NicAdamTensor * max(len(ents), 1))(*[e for e, _ in ents])
            return arr, [lr for _, lr in ents]
        if self.plan is None:
            self.table, self.table_interior = [table(grids + dec)], None
        else:
            self.table, self.table_interior = [table(boundary + dec)], table(interior)
        self._keep, self._bucket = keep, gg0.data_ptr()

    def _launch(self, tab, i, total_steps):
        arr, lrs = tab
        if not lrs:
            return
        cos = 0.5 * (1 + math.cos(math.pi * i / max(total_steps, 1)))
        for k, lr in enumerate(lrs):
            arr[k].step = i + 1
            arr[k].lr = lr * cos
        self._lib.check(self.lib.nic_adam_multi(arr, len(lrs), 0.9, 0.999, 1e-8, self._lib.stream_ptr(self.dev)), "nic_adam_multi")

    def adam_interior(self, out, i, total_steps):
        """stripe-sharded fits: Adam over the rank's interior node rows - launched between the start of the all-reduce and its completion
        (``stripe_exchange(overlap=)``): nothing it reads is touched by the exchange"""
        if self.table is None or self._bucket != out.grad_g0.data_ptr():
            self._build_table(out.grad_mlp, out.grad_g0, out.grad_g1)
        if self.table_interior is not None:
            self._launch(self.table_interior, i, total_steps)

    def adam(self, out, i, total_steps):
        """Adam (lr 0.005 decoder / 0.01 grids, image_compression.py:361-364) x CosineAnnealingLR(T_max) + the grids' clamp: one launch - or none,
        when it rode on the step's reduction (fwd_bwd(adam=)).  Stripe-sharded fits: the boundary rows and the decoder (after the exchange);
        ``adam_interior`` has run under it."""
        if self._tail_done:
            self._tail_done = False
            return
        if self.table is None or self._bucket != out.grad_g0.data_ptr():
            self._build_table(out.grad_mlp, out.grad_g0, out.grad_g1)
        for tab in self.table:
            self._launch(tab, i, total_steps)
        self._clean = True

    def n_params(self):
        return sum(p.numel() for p in self.params) + sum(g.numel() for g in self.master)


def run_sharded(args, rank, world, dev):
    """the 4K image (2D) or the video field (3D, config 4): whole-domain passes, weak or strong scaling, stripes or replicated"""
    from neural_image_compression_v2_amd import _lib, fused, models, utils
    from neural_image_compression_v2_amd.distributed import all_reduce_flat, assemble_stripes, plan_stripes, stripe_exchange
    video = args.workload == "video"
    dim, method = (3, args.method) if video else (2, 1)
    full = VID if video else (H, W)                                   # sample extents; the stripe axis is the last one
    L = full[-1]
    n_domain = int(np.prod(full))
    gdt = {"f32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}[args.grid_dtype]
    if gdt != torch.float32 and args.precision == "f32":
        raise SystemExit("16-bit grid storage needs --precision split (2D) or bf16")
    fit = Fit(dev, dim, method, grid_base=tuple(s // 4 for s in full), n_linear=args.decoder, precision=args.precision, grid_dtype=gdt)
    vworld = args.virtual_world if world == 1 and args.virtual_world > 1 else world
    strong = args.scaling == "strong"
    stripes = vworld > 1 and args.shard == "stripes"
    split_samples = vworld > 1 and (stripes or strong)               # the rank's crop is its stripe of the last sample axis
    plan = plan_stripes(L, 8, rank, vworld) if split_samples else None    # G1 cell = 8 samples at step 1/4
    if stripes:
        fit.plan = plan
    extent = tuple(full[:-1]) + ((plan.size,) if split_samples else (L,))
    passes = vworld if (stripes and not strong) else 1
    start = plan.start if split_samples else 0
    org = torch.tensor([[0] * (dim - 1) + [start]], dtype=torch.int32, device=dev)
    n_mine = int(np.prod(extent)) * passes
    n_global = n_domain if strong else n_domain * vworld
    if split_samples:
        base_mine = int(np.prod(full[:-1])) * start * passes         # global id of this rank's first sample
    else:
        base_mine = rank * n_domain                                   # replicated + weak: every rank its own pass over the whole domain
    # targets: the 4K image as an fp32 [N, 3] tensor (the reference's crop stack, built once) or read from the resident RGBX image inside the
    # step; the video field always from the resident uint8 volume (1.6 GB as an fp32 row stack)
    img = None
    if video:
        from neural_image_compression_v2_amd.sampler import rgbx_interleave
        vol = synthetic_video(dev)
        target = fused.TargetImage(rgbx_interleave(vol), 255.0, rgbx=True)
        del vol
    else:
        img = synthetic_image()
        if args.target == "image":
            from neural_image_compression_v2_amd.sampler import rgbx_interleave
            target = fused.TargetImage(rgbx_interleave(torch.round(img * 255).to(torch.uint8).to(dev)), 255.0, rgbx=True)
        else:
            sl = img[:, :, start:start + extent[-1]]
            target = sl.permute(1, 2, 0).reshape(-1, 3).repeat(passes, 1).contiguous().to(dev)
    total_steps = args.warmup + args.steps
    ev = [KernelEvents(fit.lib) for _ in range(args.steps)]
    n_small = [None]

    def step(i, events=None):
        out = fit.fwd_bwd(fit.geometry(i, extent, 1, passes, base_mine, n_global), org, target, events, adam=None if world > 1 else (i, total_steps))
        if stripes:
            if n_small[0] is None:
                n_small[0] = fused.grad_bucket_layout(fused.PathGeometry(dim, method, 0.25, 0, extent, 1), *fit.grids, n_linear=fit.nl)[0][1 + 2 * fit.nl]
            small = out.flat[:n_small[0]]
            # pack -> all-reduce (asynchronous on RCCL's stream) -> Adam over the interior node rows under it -> unpack -> (below) boundary rows + decoder
            stripe_exchange(plan, small, out.grad_g0, out.grad_g1, overlap=lambda: fit.adam_interior(out, i, total_steps),
                            **({} if world > 1 else {"reduce": lambda t, g: None}))
        else:
            all_reduce_flat(out.flat)                                 # RCCL sum of [loss | decoder grads | grid grads]
        fit.adam(out, i, total_steps)
        return out

    def kernel_only_leg(f, launches, precision=None, tgt=None):
        """per-launch HIP-event times of the fused kernel alone (no optimiser: the parameters stay put), same inputs and flags"""
        evs = [KernelEvents(f.lib) for _ in range(launches)]
        for j in range(launches + 3):
            f.fwd_bwd(f.geometry(total_steps + j, extent, 1, passes, base_mine, n_global, precision=precision), org, target if tgt is None else tgt,
                      evs[j - 3] if j >= 3 else None)
        torch.cuda.synchronize()
        return [e.elapsed_ms() for e in evs]

    # untimed: bring the clocks up before the W warm-up steps (a 45 ms timed region right after an idle start has no ramp margin)
    t_pre = time.perf_counter()
    while (time.perf_counter() - t_pre) * 1e3 < args.prewarm_ms:
        fit.fwd_bwd(fit.geometry(0, extent, 1, passes, base_mine, n_global), org, target)
        torch.cuda.synchronize()
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
        assemble_stripes(plan, *fit.master)                           # once, outside the timed region: the full grids on every rank
        if fit.mirror is not None:
            for m, p in zip(fit.mirror, fit.master):
                m.copy_(p)
    timed_kernel_ms = [e.elapsed_ms() for e in ev]                     # the fused kernel alone (nic_mark_kernel_end)
    kern_ms = float(np.mean(timed_kernel_ms))
    # the OTHER shard leg of the same strong-scaling step, same run, same timing protocol: replicated parameters, the whole gradient bucket
    # all-reduced (the north star's literal scheme; `--shard replicated` makes it the headline leg instead) - so that one line carries both
    other_leg = None
    if world > 1 and strong and stripes and not video:
        f2 = Fit(dev, dim, method, grid_base=tuple(s // 4 for s in full), n_linear=args.decoder, precision=args.precision, grid_dtype=gdt)

        def step2(i):
            o = f2.fwd_bwd(f2.geometry(i, extent, 1, 1, base_mine, n_global), org, target)
            all_reduce_flat(o.flat)
            f2.adam(o, i, total_steps)
        for i in range(3):
            step2(i)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for i in range(args.steps):
            step2(3 + i)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        tt = torch.tensor([time.perf_counter() - t2], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        e2 = float(tt.item())
        other_leg = {"shard": "replicated", "ms_per_step": round(e2 / args.steps * 1e3, 4), "value": round(n_domain * args.steps / e2 / 1e6, 2),
                     "all_reduce_bytes_per_step": int(f2.flat.numel()) * 4, "steps": args.steps, "warmup": 3}
        del f2
    psnr = None
    if not video:
        # the metric's "+ PSNR" (outside the timed region): decode the whole image with the current parameters, PSNR with peak 2^8
        # against the synthetic target (utils.py:117-130) - after warmup + steps optimiser steps from a random initialisation
        dgeo = fused.PathGeometry(dim=2, method=1, step_number=0.25, mip_level=0, extent=(H, W), num_crops=1,
                                  flags=_lib.NIC_FLAG_ORIGINS_ALIGNED, split_bf16=args.precision != "f32")
        rec = fused.fused_forward(dgeo, fit.master[0], fit.master[1], torch.zeros(1, 2, dtype=torch.int32, device=dev), fit.params)
        ref_img = img.permute(1, 2, 0).reshape(-1, 3).contiguous().to(dev)
        psnr = float(utils.calculate_psnr(models.quantize_to_bit(rec, 8), models.quantize_to_bit(ref_img, 8)))
        del rec, ref_img
    # statistics + the other arithmetic modes, after the timed region (every rank runs them: same work everywhere, no collectives)
    extra = {}
    if args.stat_launches > 0:
        stat_main = kernel_only_leg(fit, args.stat_launches)
        if fit.mirror is None and fit.nl == 3 and vworld == 1:
            for other in ("split", "f32", "bf16", "fp16"):
                if other != args.precision:
                    extra[other] = kernel_only_leg(fit, max(args.stat_launches // 2, 10), precision=other)
    else:
        stat_main = None
    # the north star's literal configuration as a WHOLE step (2D headline runs only): 5-Linear "4 x 64" decoder, bf16 grid storage + fp32
    # masters, plain-bf16 products; fused step + reduce + Adam over all 12 tensors with the 16-bit mirrors rewritten.  Beside it the same
    # decoder on the split-bf16 kernel (fp32 grids), kernel only.
    lit = None
    if args.stat_launches > 0 and not video and vworld == 1 and not (args.decoder == 5 and args.precision == "bf16" and gdt == torch.bfloat16):
        f5 = Fit(dev, 2, 1, grid_base=(H // 4, W // 4), n_linear=5, precision="bf16", grid_dtype=torch.bfloat16, seed=1)
        k5, w5 = max(args.stat_launches // 4, 10), 3
        ev5 = [KernelEvents(f5.lib) for _ in range(k5)]

        def step5(i, e=None):
            o = f5.fwd_bwd(f5.geometry(i, extent, 1, passes, base_mine, n_global), org, target, e, adam=None if world > 1 else (i, w5 + k5))
            f5.adam(o, i, w5 + k5)
            return o
        for i in range(w5):
            step5(i)
        torch.cuda.synchronize()
        t5 = time.perf_counter()
        for i in range(k5):
            o5 = step5(w5 + i, ev5[i])
        torch.cuda.synchronize()
        dt5 = (time.perf_counter() - t5) / k5
        km5 = [e.elapsed_ms() for e in ev5]
        f5s = Fit(dev, 2, 1, grid_base=(H // 4, W // 4), n_linear=5, precision="split", seed=1)
        ks5 = kernel_only_leg(f5s, max(args.stat_launches // 6, 10))
        lit = (dt5, km5, float(o5.loss), ks5, k5, w5)
        del f5s

    if rank == 0:
        unit = "Mvoxels/s" if video else "Mpixels/s"
        done = n_domain * (1 if strong else world) * args.steps       # samples the whole job processed in the timed region
        rate = done / elapsed / 1e6
        grids_txt = " + ".join(str(list(g.shape)) for g in fit.master)
        if stripes:
            xb = 4 * (n_small[0] + (vworld - 1) * 12 * sum(int(g[0, 0].numel()) for g in fit.master))
            par = (f"dp{vworld}, {'strong' if strong else 'weak'} scaling, grids sharded in stripes of the last sample axis ({plan.size} samples x {passes} "
                   f"pass{'es' if passes > 1 else ''} per rank and step, Adam over the rank's own node rows); exchange per step = loss + decoder grads + "
                   f"{vworld - 1} boundary node rows of G0 and G1 = {xb} B all-reduced" + (" (virtual: one process, no collectives)" if world == 1 else ""))
        elif vworld > 1:
            par = (f"dp{world}, {'strong' if strong else 'weak'} scaling, replicated parameters, "
                   f"{'the rank takes its stripe of the samples' if strong else 'every rank covers the whole domain'}; {4 * fit.flat.numel()} B all-reduced per step")
        else:
            par = "dp1"
        traffic = None
        tp = os.path.join(ROOT, "profiles", "traffic.json")            # HBM bytes per launch from the rocprofv3 --pmc passes, if collected
        if os.path.exists(tp) and not video and args.precision == "split" and fit.nl == 3 and fit.mirror is None:
            try:
                traffic = json.load(open(tp)).get("fused_kernel_bytes_per_launch")
            except Exception:
                traffic = None
        gb = 4 if fit.mirror is None else 2
        dec_txt = "3xLinear(64) GELU decoder" if fit.nl == 3 else "5xLinear(64) GELU decoder (the north star's \"4 x 64\")"
        if video:
            wl = (f"1920x1080x64 video field as a 3D fit (BASELINE config 4), every voxel once per step and replica: dense G0 + G1 {grids_txt}, method {method} "
                  f"({'tetrahedral G0, sinusoidal PE' if method == 4 else '8 raw G0 corners, triangular PE'}), {dec_txt}, targets from the resident RGBX volume, "
                  "in-kernel Threefry-4x32-12 noise, MSE, fused fwd+bwd + gradient exchange + Adam + clamp")
            metric = "Mvoxels/sec train-step (fwd+bwd), 1920x1080x64 video field, 1/2/4/8 MI355X"
        else:
            wl = (f"3840x2160 RGB fit, every pixel once per step: dense G0 [12,961,541] + G1 [12,481,271] grid pair (reference semantics, no-mip"
                  f"{', ' + args.grid_dtype + ' storage with fp32 masters' if fit.mirror is not None else ''}), triangular PE, {dec_txt}, "
                  "in-kernel Threefry-4x32-12 noise, MSE, fused fwd+bwd + grad all-reduce + Adam + clamp")
            metric = "Mpixels/sec train-step (fwd+bwd) + PSNR, 4K RGB, 1/2/4/8 MI355X"
        res = {
            "metric": metric, "value": round(rate, 2), "unit": unit, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": DTYPE_NAME[args.precision], "data": "synthetic",
            "config": {"workload": wl, "samples_per_step_per_gpu": n_mine, "parallelism": par, "final_loss": round(loss, 6)},
        }
        if psnr is not None:
            res["config"]["psnr_db_after_these_steps"] = round(psnr, 3)
        if other_leg is not None:
            res["shard_legs"] = {"stripes": {"ms_per_step": res["ms_per_step"], "value": res["value"]}, "replicated": other_leg,
                                 "note": "`value` is the stripes leg; both legs ran in this process group, back to back"}
        nodes = [int(g[0].numel()) for g in fit.master]
        res["roofline"] = roofline_record(dim, method, args.precision, fit.nl, kern_ms, n_mine, traffic,
                                          {"timed_steps": pct(timed_kernel_ms), **({"stat_leg": pct(stat_main)} if stat_main else {})}, gb, nodes)
        for other, xs in extra.items():
            # another arithmetic mode on the same inputs in the same run: its median launch time carries the record
            r = roofline_record(dim, method, other, fit.nl, float(np.median(xs)), n_mine, None, {"stat_leg": pct(xs)}, gb, nodes)
            r["m%s_s_kernel_only" % ("vox" if video else "pix")] = round(n_mine / float(np.median(xs)) / 1e3, 1)
            if other == "bf16":
                r["parity"] = "against the precision-emulating oracle at 1e-3 (tests/test_gpu_bf16.py); ~5e-3 from fp32 arithmetic"
            res["roofline_" + other] = r
        if lit is not None:
            dt5, km5, loss5, ks5, k5, w5 = lit
            r = roofline_record(2, 1, "bf16", 5, float(np.median(km5)), n_mine, None, {"timed_steps": pct(km5)}, 2)
            r.update({"decoder": "Linear(73,64) + 3 x Linear(64,64) + Linear(64,3), GELU, Sigmoid (n_linear = 5)",
                      "grids": "bfloat16 storage (966 B per sample), fp32 masters + Adam state", "whole_step": True,
                      "ms_per_step": round(dt5 * 1e3, 4), "steps": k5, "warmup": w5, "mpix_s": round(n_mine / dt5 / 1e6, 1),
                      "step": "fused_q16_kernel + reduce_q16_kernel + nic_adam_multi over 10 decoder tensors + 2 grids (16-bit mirrors rewritten)",
                      "final_loss": round(loss5, 6),
                      "split_bf16_products_fp32_grids_kernel_only": {"kernel": kernel_name(2, 1, "split", 5), "kernel_ms": round(float(np.median(ks5)), 4),
                                                                     "mpix_s": round(n_mine / float(np.median(ks5)) / 1e3, 1), "stats": pct(ks5)}})
            res["roofline_4x64"] = r
        if world == 1 and not args.no_cpu_baseline and not video:
            res["cpu_baseline"] = cpu_baseline(img)
        print(json.dumps(res), flush=True)


def run_fits64(args, rank, world, dev):
    """BASELINE config 5: 64 independent 1080p fits, 64 / N per GPU, one step of each per sweep, back to back (a persistent
    workgroup keeps its CU's LDS for the whole launch, so side-by-side fits only divide the chip).  Replicas only - no collective."""
    from neural_image_compression_v2_amd import _lib
    HH, WW, NF = 1080, 1920, 64
    lo = rank * NF // world
    hi = (rank + 1) * NF // world
    g = torch.Generator(device="cpu").manual_seed(99)
    fits, targets = [], []
    for k in range(NF):
        t = torch.rand(HH * WW, 3, generator=g)                       # every rank draws the same 64 targets and keeps its own
        if lo <= k < hi:
            fits.append(Fit(dev, 2, 1, grid_base=(HH // 4, WW // 4), n_linear=args.decoder, precision=args.precision, seed=k))
            targets.append(t.to(dev))
    org = torch.zeros(1, 2, dtype=torch.int32, device=dev)
    total = args.warmup + args.steps

    def sweep(i):
        outs = []
        for f, t in zip(fits, targets):
            o = f.fwd_bwd(f.geometry(i, (HH, WW)), org, t, adam=(i, total))
            f.adam(o, i, total)
            outs.append(o)
        return outs
    for i in range(args.warmup):
        sweep(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        outs = sweep(args.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if rank == 0:
        rate = NF * HH * WW * args.steps / elapsed / 1e6
        _, fl, byt = work_per_sample(2, 1, args.decoder)
        print(json.dumps({
            "metric": "aggregate Mpixels/sec, 64 independent 1080p fits (BASELINE config 5), 1/2/4/8 MI355X", "value": round(rate, 2), "unit": "Mpixels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": DTYPE_NAME[args.precision], "data": "synthetic",
            "config": {"workload": f"64 x (1920 x 1080 fit, own grids [12,481,271] + [12,241,136], own {args.decoder}-Linear decoder, own Adam state); a step = one "
                                   f"training step of every fit, {hi - lo} fits per GPU back to back on the whole chip each",
                       "parallelism": f"replicas only: rank r owns fits [{NF}r/{world}, {NF}(r+1)/{world}), no collective", "loss_fit0": round(float(outs[0].loss), 6)},
            "roofline": {"bound": "hbm", "achieved": round(byt * rate * 1e6 / world / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": round(byt * rate * 1e6 / world / 1e9 / PEAK_HBM_GBS, 4), "traffic": None, "bytes_per_sample": byt,
                         "note": "per GPU, whole sweep (fused kernel + reduce + Adam of every fit) over the algorithmic bytes"}}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stat-launches", type=int, default=120, help="launches of the post-run statistics leg (median / p10 / p90 of the kernel time); 0: skip the extra legs")
    ap.add_argument("--prewarm-ms", type=float, default=300.0, help="untimed kernel launches before the W warm-up steps (clock ramp)")
    ap.add_argument("--precision", choices=["split", "f32", "bf16", "fp16"], default="split",
                    help="split: every matrix product of the step as hi + lo bf16 pairs on the bf16 matrix pipe, fp32 accumulate (gradients "
                         "within 5e-6 of the fp32 kernel; the product's default); f32: v_mfma_f32_32x32x2_f32 throughout; bf16: plain bf16 products")
    ap.add_argument("--decoder", type=int, choices=[3, 5], default=3, help="Linear layers of the decoder: 3 (the reference) or 5 (the north star's \"4 x 64\")")
    ap.add_argument("--grid-dtype", choices=["f32", "bf16", "fp16"], default="f32", help="grid STORAGE the kernels gather from (fp32 masters + Adam state either way)")
    ap.add_argument("--target", choices=["tensor", "image"], default="tensor",
                    help="4k: tensor = resident fp32 [N,3] targets (the reference's crop stack, built once); image = targets read from the "
                         "resident RGBX uint8 image inside the step (a third of the bytes, one dword load per sample)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong",
                    help="N > 1.  weak: the whole domain per RANK and step; strong: ONE pass over the domain per step, split N ways")
    ap.add_argument("--shard", choices=["stripes", "replicated"], default="stripes",
                    help="N > 1.  stripes: every rank owns a stripe of the last sample axis (a contiguous block of grid node rows); the step "
                         "exchanges the loss, the decoder gradients and one boundary node row per neighbour pair.  replicated: replicated "
                         "parameters, the whole gradient bucket is all-reduced")
    ap.add_argument("--method", type=int, choices=[3, 4], default=4, help="--workload video: COMPRESSION_METHOD")
    ap.add_argument("--graph", type=int, default=0, help="--workload default / default3d: time replays of a captured hipGraph of this many training steps "
                                                         "(ImageCompression.train_models_graph) instead of the host loop")
    ap.add_argument("--virtual-world", type=int, default=0,
                    help="diagnostic, one process: run rank 0's share of an N-rank stripe-sharded step without the collectives")
    ap.add_argument("--workload", default="4k", choices=["4k", "video", "fits64", "lut33", "vol64", "vol128", "slab", "default", "default3d", "fits8", "multilevel"],
                    help="4k (default): BASELINE configs[1], the headline; video / fits64: configs 4 / 5 (any --gpus); the others: bench_workloads.py "
                         "(one GPU, same JSON shape - records for profiles/; the driver runs the default only)")
    ap.add_argument("--launch-check", action="store_true",
                    help="plumbing check of the N-rank launch (tests, no GPU): ranks rendezvous over gloo, sum their ranks and - for the sharded "
                         "workloads - print the per-rank plan; rank 0 prints a JSON line")
    args = ap.parse_args()

    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    # no cyclic-GC pauses inside timed regions: with torch imported a generation-2 collection takes 40-50 ms of host time - invisible behind 2 ms steps
    # (the GPU has work queued), a 3 x error on a 40-step loop of 0.4 ms steps (measured at call 41 of `--workload vol128`; ab/dbg/stall_dbg.py).  The
    # process is short-lived and its steady state allocates no cycles worth collecting.
    import gc
    gc.collect()
    gc.freeze()
    gc.disable()

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        print(f"bench.py: WORLD_SIZE={world} but --gpus {args.gpus}: launch with --nproc-per-node {args.gpus}", file=sys.stderr)
        sys.exit(2)
    if args.launch_check:
        dist.init_process_group("gloo")
        t = torch.tensor([float(rank)])
        dist.all_reduce(t)
        rec = {"launch_check": True, "n_gpus": dist.get_world_size(), "rank_sum": float(t.item()), "workload": args.workload, "scaling": args.scaling}
        if args.workload in ("4k", "video"):
            from neural_image_compression_v2_amd.distributed import plan_stripes
            L = VID[-1] if args.workload == "video" else W
            plans = [plan_stripes(L, 8, r, world) for r in range(world)] if world > 1 else []
            rec["stripes"] = [[p.start, p.size] for p in plans]
            rec["covered"] = sum(p.size for p in plans) if plans else L
        elif args.workload == "fits64":
            rec["fits_per_rank"] = [(r + 1) * 64 // world - r * 64 // world for r in range(world)]
        if rank == 0:
            print(json.dumps(rec), flush=True)
        dist.destroy_process_group()
        return

    if args.workload not in ("4k", "video", "fits64"):
        if world != 1:
            ap.error(f"--workload {args.workload} runs on one GPU")
        import bench_workloads
        bench_workloads.run(args)
        return
    local = int(os.environ.get("LOCAL_RANK", 0)) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        backend = os.environ.get("NIC_DIST_BACKEND", "nccl")         # "gloo": rehearsal of the N > 1 path on a box with one GPU
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
        assert dist.get_world_size() == args.gpus
    if args.workload == "fits64":
        run_fits64(args, rank, world, dev)
    else:
        run_sharded(args, rank, world, dev)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
