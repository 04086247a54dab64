"""Device-side sampler and resident RGBX target pyramid.

The reference draws a step's LOD with Python's ``random`` and its crop origins with ``torch.randint`` on the host, slices the
crops out of per-LOD fp32 dataset tensors and stacks them (image_compression.py:26-50, 429-477).  The default training loop
of this package keeps those host RNG calls (so a fit replays the reference's stream draw for draw); this module is the opt-in
replacement that keeps the host out of the step:

* :class:`DeviceSampler` - a counter-based generator (Threefry-4x32-12 keyed by seed, step, crop; ``nic_sampler_*``,
  csrc/nic_device.hpp::sampler_block): the LOD - which fixes the launch geometry - is a pure host function of (seed, step),
  the crop origins are written by a tiny kernel straight into the device buffer the fused step reads.  No host RNG state,
  no upload per step, same laws as the reference's draws.
* :func:`build_rgbx_pyramid` - the dataset as interleaved uint8 RGBX levels (one dword per sample) built once on the device:
  level 0 from the image's 8-bit codes, level k + 1 by a 2 x 2 box filter of level k (2D).  3D volumes have one level per LOD
  pointing at the same tensor, like the reference (image_compression.py:470-477).
"""
from __future__ import annotations

import ctypes
from typing import List, Sequence

import torch

from . import _lib, fused


def rgbx_interleave(image_u8: torch.Tensor) -> torch.Tensor:
    """planar uint8 ``[3, S0, S1(, S2)]`` -> int32 ``[S0, S1(, S2)]``, R | G << 8 | B << 16"""
    if image_u8.dtype != torch.uint8 or image_u8.shape[0] != 3:
        raise ValueError("expects the uint8 codes [3, ...]")
    if not image_u8.is_cuda:
        raise RuntimeError(f"image lives on {image_u8.device}: this package only runs on a HIP device (no CPU path)")
    src = image_u8.contiguous()
    n = src[0].numel()
    out = torch.empty(tuple(src.shape[1:]), dtype=torch.int32, device=src.device)
    with torch.cuda.device(src.device):
        _lib.check(_lib.load().nic_rgbx_interleave(_lib.ptr(src), n, _lib.ptr(out), _lib.stream_ptr(src.device)), "nic_rgbx_interleave")
    return out


def rgbx_downsample2(level: torch.Tensor) -> torch.Tensor:
    """next level of a 2D RGBX mip chain: 2 x 2 box filter, round to nearest"""
    if level.dtype != torch.int32 or level.dim() != 2 or not level.is_cuda:
        raise ValueError("expects a 2D RGBX level on the device")
    s0, s1 = int(level.shape[0]), int(level.shape[1])
    out = torch.empty(s0 // 2, s1 // 2, dtype=torch.int32, device=level.device)
    with torch.cuda.device(level.device):
        _lib.check(_lib.load().nic_rgbx_downsample2(_lib.ptr(level.contiguous()), s0, s1, _lib.ptr(out), _lib.stream_ptr(level.device)),
                   "nic_rgbx_downsample2")
    return out


def resize_coeffs(in_size: int, out_size: int):
    """Pillow's ``precompute_coeffs`` + ``normalize_coeffs_8bpc`` for the BILINEAR filter (src/libImaging/Resample.c) - the arithmetic behind
    ``transforms.Resize`` on a PIL image (image_compression.py:434-440) - restated operation for operation in double precision: per output index
    the first tap, the tap count, and the taps' weights as int32 in 2^-22 units.  Returns ``(bounds [out, 2], kk [out, ksize], ksize)`` (numpy)."""
    import math
    import numpy as np
    scale = float(in_size) / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = 1.0 * filterscale                                       # bilinear: support 1
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = []
        ww = 0.0
        for x in range(xmax):
            a = (x + xmin - center + 0.5) * ss
            if a < 0.0:
                a = -a
            w = 1.0 - a if a < 1.0 else 0.0
            k.append(w)
            ww += w
        for x in range(xmax):
            v = k[x] / ww if ww != 0.0 else k[x]
            kk[xx, x] = int(-0.5 + v * (1 << 22)) if v < 0 else int(0.5 + v * (1 << 22))
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


def rgbx_resize(level0: torch.Tensor, out0: int, out1: int) -> torch.Tensor:
    """Pillow's BILINEAR ``Image.resize`` of an RGBX image ``[S0, S1]`` to ``[out0, out1]`` (``transforms.Resize((out0, out1))`` of the reference,
    image_compression.py:434-440), bit-exact: the pass along the contiguous axis first, then across rows, each rounding to bytes"""
    if level0.dtype != torch.int32 or level0.dim() != 2 or not level0.is_cuda:
        raise ValueError("expects a 2D RGBX level on the device")
    s0, s1 = int(level0.shape[0]), int(level0.shape[1])
    dev = level0.device
    lib = _lib.load()
    cur = level0.contiguous()
    with torch.cuda.device(dev):
        for axis, out in ((1, out1), (0, out0)):
            in_size = s1 if axis == 1 else s0
            if out == in_size:
                continue                                              # Pillow skips a pass that does not change the size
            bounds, kk, ksize = resize_coeffs(in_size, out)
            b, k = torch.from_numpy(bounds).to(dev), torch.from_numpy(kk).to(dev)
            dst = torch.empty((s0, out) if axis == 1 else (out, s1), dtype=torch.int32, device=dev)
            _lib.check(lib.nic_rgbx_resample_axis(_lib.ptr(cur), s0, s1, axis, out, _lib.ptr(b), _lib.ptr(k), ksize, _lib.ptr(dst), _lib.stream_ptr(dev)),
                       "nic_rgbx_resample_axis")
            cur = dst
            s0, s1 = int(cur.shape[0]), int(cur.shape[1])
    return cur


def build_rgbx_pyramid(image_u8: torch.Tensor, levels: int, den: float = 255.0, mip_filter: str = "resize") -> List[fused.TargetImage]:
    """``levels`` resident target images (LOD 0 .. levels - 1) as :class:`fused.TargetImage` in the RGBX layout.  2D, ``mip_filter``:
    "resize" (default) = the reference's own chain - every level is ``transforms.Resize(S // 2^i)`` of the ORIGINAL image, i.e. Pillow's BILINEAR
    resize (image_compression.py:432-440), reproduced bit for bit by :func:`rgbx_resize`; "box" = level k + 1 by a 2 x 2 box filter of level k
    (round 2's filter).  3D: every LOD reads the full-resolution volume, like the reference (:470-477)."""
    if mip_filter not in ("resize", "box"):
        raise ValueError("mip_filter is 'resize' (the reference's) or 'box'")
    base = rgbx_interleave(image_u8)
    cur = base
    out = [fused.TargetImage(cur, den, rgbx=True)]
    for i in range(1, levels):
        if base.dim() == 2:
            cur = rgbx_resize(base, int(base.shape[0]) // 2 ** i, int(base.shape[1]) // 2 ** i) if mip_filter == "resize" else rgbx_downsample2(cur)
        out.append(fused.TargetImage(cur, den, rgbx=True))
    return out


class DeviceSampler:
    """``draw(step, uniform_distribution, max_mip_level, crop_size, data_sizes, num_crops, dim)`` -> ``(origins, lod)`` with
    ``origins`` an int32 ``[num_crops, dim]`` DEVICE tensor (valid on the current stream) and ``lod`` a host int."""

    def __init__(self, seed: int, device, max_crops: int = 64):
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("DeviceSampler needs a HIP device")
        self._buf = torch.empty(max_crops * 3, dtype=torch.int32, device=self.device)

    def lod(self, step: int, uniform_distribution: bool, max_mip_level: int) -> int:
        rc = _lib.load().nic_sampler_lod_host(self.seed, int(step), 1 if uniform_distribution else 0, int(max_mip_level))
        if rc < 0:
            _lib.check(rc, "nic_sampler_lod_host")
        return int(rc)

    def draw(self, step: int, uniform_distribution: bool, max_mip_level: int, crop_size: int, data_sizes: Sequence[int], num_crops: int, dim: int):
        lod = self.lod(step, uniform_distribution, max_mip_level)
        re_crop = max(1, crop_size // (2 ** lod))                                  # image_compression.py:38
        rng = int(data_sizes[lod]) - re_crop + 1                                   # torch.randint(0, data_size - re_crop + 1)
        if num_crops * dim > self._buf.numel():
            self._buf = torch.empty(num_crops * dim, dtype=torch.int32, device=self.device)
        out = self._buf[:num_crops * dim].view(num_crops, dim)
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().nic_sampler_draw_origins(self.seed, int(step), num_crops, dim, rng, _lib.ptr(out), _lib.stream_ptr(self.device)),
                       "nic_sampler_draw_origins")
        return out, lod

    def origins_host(self, step: int, num_crops: int, dim: int, rng: int) -> torch.Tensor:
        """the same draw evaluated on the host (tests, debugging)"""
        arr = (ctypes.c_int32 * (num_crops * dim))()
        _lib.check(_lib.load().nic_sampler_origins_host(self.seed, int(step), num_crops, dim, int(rng), arr), "nic_sampler_origins_host")
        return torch.tensor(list(arr), dtype=torch.int32).view(num_crops, dim)
