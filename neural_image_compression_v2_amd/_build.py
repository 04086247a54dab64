"""Builds libnicv2_hip.so (the C-ABI library, include/nicv2_hip.h) in-tree with hipcc for gfx950.

    python -m neural_image_compression_v2_amd._build [--force]

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the GPU box with the snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libnicv2_hip.so")
SOURCES = ["simple_kernels.hip", "decoder_general.hip", "fused_capi.hip", "fused_m1.hip", "fused_m2.hip", "fused_m3.hip", "fused_m4.hip", "fused_t16.hip", "fused_mlpn.hip", "fused_q1.hip", "fused_q2.hip", "fused_q3.hip", "fused_q4.hip"]
# non-default FEATURE_PYRAMID_CHANNELS / PE_CHANNELS on the plain-bf16 kernels: (layout, C, P), one translation unit each
SOURCES += [f"fused_qc_{l}_{c}_{p}.hip" for l, c, p in [(1, 4, 6), (1, 8, 6), (1, 16, 6), (1, 12, 4), (1, 12, 8), (2, 4, 6), (2, 8, 6), (2, 16, 6), (2, 12, 4), (2, 12, 8),
                                                         (3, 4, 6), (3, 8, 6), (4, 4, 6), (4, 8, 6), (4, 16, 6)]]
# multi-level layouts (fused_q16.hpp::QML): (levels, C, n_linear), one translation unit each
ML_LIST = [(2, 4, 3), (3, 4, 3), (5, 4, 3), (2, 4, 5), (3, 4, 5), (2, 12, 3), (3, 12, 3)]
SOURCES += [f"fused_ml_{l}_{c}_{n}.hip" for l, c, n in ML_LIST]
HEADERS = ["nic_device.hpp", "nic_adam.hpp", "fused_kernel.hpp", "fused_launch.hpp", "fused_train16.hpp", "fused_t16.hpp", "fused_mlpn.hpp", "fused_q16.hpp", "fused_q16_launch.hpp", os.path.join("..", "..", "include", "nicv2_hip.h")]
# -amdgpu-mfma-vgpr-form: MFMA results that vector instructions consume may live in the architectural VGPRs instead of bouncing
# through v_accvgpr_read / write (split training kernel: 656 -> 423 of them, -0.7 %; fp32 2D 18 -> 0 spills; 3D 170 -> 115 / 135 -> 85)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-mllvm", "-amdgpu-mfma-vgpr-form",
         "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + hdrs):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        extra = ["-ffp-contract=off"] if os.path.basename(s) == "simple_kernels.hip" else []   # op-by-op rounding like eager torch
        # the plain-bf16 kernels without SLP vectorisation (the guide's anti-lever: adjacent scalar f32 operations packed into v_pk_* beside MFMAs):
        # 4K launch 1.420 -> 1.401 ms, 128^3 method 3 0.483 -> 0.461, method 4 0.426 -> 0.417 (interleaved A/B, identical results); the split and fp32
        # kernels do not move (or lose 1 %): they keep the default
        # (fused_t16: 2.122 -> 2.104 ms on the final kernel; it did not move before the 16x16x16 products went in)
        if os.path.basename(s).startswith(("fused_q", "fused_ml_")) or os.path.basename(s) == "fused_t16.hip":
            extra.append("-fno-slp-vectorize")
        cmd = [hipcc, *FLAGS, *extra, "-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s}:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[build] {os.path.basename(s)}", flush=True)
        return o

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(cc, jobs))
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[build] linked {LIB}", flush=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
