"""Feature-pyramid grids: allocation, level maps, corner gathers, G0/G1 assembly, clamp / quantise / codec -
the call surface of the reference's Projects/fp_def.py, executed by the HIP library.

A pyramid is a Python list of leaf tensors [C, (Z,) Y, X] exactly like the reference's, so
``optim.Adam([{'params': fp, ...}])``, in-place clamping and list rebinding after the freeze keep working
(image_compression.py:227-231, 361-364).  The kernels read the tensors by pointer on every call.
"""
from __future__ import annotations

import ctypes
from collections import defaultdict

import torch

from . import _lib, fused
from .models import load4fp, quantize4fp, save4fp


def return_2_power(base_size):
    """floor(log2) by halving (fp_def.py:8-14)"""
    count, x = 0, int(base_size)
    while x != 1:
        x //= 2
        count += 1
    return count


def return_pyramid_levels(base_size):
    """number of (G0, G1) pairs of a full mip pyramid (fp_def.py:18-20)"""
    return (return_2_power(base_size) + 1) // 2


def create_pyramid_mip_levels(image_size, base_size):
    """mip level -> feature level: clamp(mip // 2 - 1, 0, levels - 1) (fp_def.py:24-34)"""
    levels = return_pyramid_levels(base_size)
    d = defaultdict(int)
    for mip in range(return_2_power(image_size) + 1):
        d[mip] = min(max(mip // 2 - 1, 0), levels - 1)
    return d


def _q_range(num_bits):
    return -(pow(2, num_bits) - 1) / pow(2, num_bits + 1), 1 / 2


def _create(base, dim, channels, num_bits, device, dtype, no_mip):
    """``dtype`` float32: the reference's fp32 leaves.  float16 (what the reference's MLP_NUM_DTYPE = 16 maps its grids to, utils.py:301-313;
    its own 16-bit run does not train, readme.md:9) or bfloat16: 16-bit grid STORAGE - every returned leaf is the fp32 MASTER the optimiser
    updates (same init expression, rounded to the storage type so that master == stored value at the start) and carries the 16-bit tensor
    the kernels gather from as ``leaf.mirror16``; the fused entry points pick the mirror up by themselves (fused.grid_storage),
    ``optim.FusedAdam.set_mirror`` keeps it in step with the master."""
    if dtype not in (None, torch.float32, torch.float16, torch.bfloat16):
        raise NotImplementedError("grids are fp32, or float16 / bfloat16 storage with fp32 masters")
    base = (int(base),) * dim if isinstance(base, int) else tuple(int(b) for b in base)
    if len(base) != dim:
        raise ValueError("one base size per axis")
    levels = 1 if no_mip else return_pyramid_levels(min(base))
    lo, hi = _q_range(num_bits)
    pyramid = []
    for i in range(levels * 2):
        shape = [channels] + [b // (2 ** i) + 1 for b in reversed(base)]     # tensor axes (z,) y, x
        # the reference's own expression (fp_def.py:54,76): same RNG stream and rounding as it for a given seed / device
        g = (hi - lo) * torch.rand(*shape, device=device, dtype=torch.float32) + lo
        if dtype in (torch.float16, torch.bfloat16):
            mirror = g.to(dtype)
            g = mirror.to(torch.float32)
            g.requires_grad_(True)
            g.mirror16 = mirror
        else:
            g.requires_grad_(True)
        pyramid.append(g)
    return pyramid, levels


def create_pyramid(base_size, channels, num_bits, device, dtype=torch.float32, no_mip=False):
    """2*levels grids [C, S_i+1, S_i+1], S_i = base // 2^i, U[q_min, 1/2] init (fp_def.py:37-56).  ``base_size`` may
    also be a per-axis tuple (x, y): non-square images reduce to the reference on squares."""
    return _create(base_size, 2, channels, num_bits, device, dtype, no_mip)


def create_pyramid_3d(base_size, channels, num_bits, device, dtype=torch.float32, no_mip=False):
    """fp_def.py:59-78"""
    return _create(base_size, 3, channels, num_bits, device, dtype, no_mip)


def _gather(grid, corner_set, xi, yi, zi=None):
    g = _lib.require_cuda_f32(grid.detach(), "grid")
    C = g.shape[0]
    nx, ny, nz = g.shape[-1], g.shape[-2], (g.shape[-3] if g.dim() == 4 else 1)
    idx = [t.to(torch.int32).contiguous() for t in ((xi, yi) if zi is None else (xi, yi, zi))]
    n = idx[0].numel()
    K = 8 if corner_set == 1 else 4
    out = torch.empty(K, C, n, dtype=torch.float32, device=g.device)
    with torch.cuda.device(g.device):
        _lib.check(_lib.load().nic_gather_corners(_lib.ptr(g), C, nx, ny, nz, _lib.ptr(idx[0]), _lib.ptr(idx[1]),
                                                  _lib.ptr(idx[2] if zi is not None else None), n, corner_set, _lib.ptr(out),
                                                  _lib.stream_ptr(g.device)), "nic_gather_corners")
    return tuple(out[k] for k in range(K))


def create_g(fp, fl, j, x_indices, y_indices):
    """4 corners of grid fp[2 fl + j]: (y,x), (y+1,x), (y,x+1), (y+1,x+1) (fp_def.py:81-86)"""
    return _gather(fp[fl * 2 + j], 0, x_indices, y_indices)


def create_g_3d(fp, fl, j, x_indices, y_indices, z_indices):
    """8 corners, z fastest (fp_def.py:89-104)"""
    return _gather(fp[fl * 2 + j], 1, x_indices, y_indices, z_indices)


def create_g_3d_v2(fp, fl, j, x_indices, y_indices, z_indices):
    """4 tetrahedral corners (fp_def.py:107-112)"""
    return _gather(fp[fl * 2 + j], 2, x_indices, y_indices, z_indices)


def _range_origin(r, name):
    """the reference always passes arange(sample_number) (image_compression.py:84-85,116-118); a unit-stride
    range starting anywhere is accepted and folded into the origin"""
    n = int(r.numel())
    first, last = (float(v) for v in r[[0, -1]].tolist())
    if first != int(first) or last - first != n - 1:
        raise NotImplementedError(f"{name} must be a unit-stride integer range")
    return int(first), n


def _g0_g1(fp, fl, origin, step_number, ranges, pe_channels, method, use_tri_pe):
    dim = len(origin)
    offs, ext = zip(*[_range_origin(r, "range") for r in ranges])
    org = [int(o) + f for o, f in zip(origin, offs)]
    g0, g1 = fp[fl * 2], fp[fl * 2 + 1]
    geo = fused.PathGeometry(dim=dim, method=method, step_number=step_number, mip_level=0.0, extent=tuple(ext), num_crops=1,
                             channels=g0.shape[0], pe_channels=pe_channels, use_tri_pe=use_tri_pe)
    rows = fused.encode_split(geo, g0, g1, [org])
    C = g0.shape[0]
    k0 = 4 if (dim == 2 or method == 4) else 8
    k1 = 4 if dim == 2 else 8
    parts = [rows[i * C:(i + 1) * C] for i in range(k0 + k1)]
    return (*parts, rows[(k0 + k1) * C:])


def create_g0_g1(fp, fl, x, y, step_number, x_range, y_range, pe_channels, device=None, dtype=None, use_tri_pe=True):
    """9-tuple: 4 raw G0 corners [C, n], 4 bilinearly weighted G1 corners, PE [2P, n]; samples x-outer, y-inner
    (fp_def.py:115-145)"""
    return _g0_g1(fp, fl, (x, y), step_number, (x_range, y_range), pe_channels, 1, use_tri_pe)


def create_g0_g1_3d(fp, fl, x, y, z, step_number, x_range, y_range, z_range, pe_channels, device=None, dtype=None):
    """17-tuple: 8 G0 corners, 8 G1 corners with the reference's (permuted) weights, triangular PE [3P, n]
    (fp_def.py:148-184)"""
    return _g0_g1(fp, fl, (x, y, z), step_number, (x_range, y_range, z_range), pe_channels, 3, True)


def create_g0_g1_3d_v2(fp, fl, x, y, z, step_number, x_range, y_range, z_range, pe_channels, device=None, dtype=None):
    """13-tuple: 4 tetrahedral G0 corners, 8 weighted G1 corners, sinusoidal PE (fp_def.py:187-223)"""
    return _g0_g1(fp, fl, (x, y, z), step_number, (x_range, y_range, z_range), pe_channels, 4, False)


def fp_quantize_clamp(fp, fl, num_bits):
    """in-place clamp of the pair just trained to [q_min, 1/2] (fp_def.py:227-232)"""
    lo, hi = _q_range(num_bits)
    lib = _lib.load()
    with torch.no_grad():
        for g in (fp[fl * 2], fp[fl * 2 + 1]):
            t = _lib.require_cuda_f32(g, "grid")
            if t.data_ptr() != g.data_ptr():
                raise RuntimeError("grids must be contiguous")
            with torch.cuda.device(t.device):
                _lib.check(lib.nic_clamp(_lib.ptr(t), t.numel(), lo, hi, _lib.stream_ptr(t.device)), "nic_clamp")


def fp_quantize(fp, fl, num_bits):
    """fp_def.py:236-239"""
    with torch.no_grad():
        fp[fl * 2] = quantize4fp(fp[fl * 2], num_bits)
        fp[fl * 2 + 1] = quantize4fp(fp[fl * 2 + 1], num_bits)


def fp_all_quantize(fp, num_bits):
    """new list of quantised copies (fp_def.py:242-247)"""
    return [quantize4fp(g, num_bits) for g in fp]


def fp_savable(fp, num_bits, dtype=torch.uint8):
    """list of uint8 tensors for torch.save (fp_def.py:250-255)"""
    return [save4fp(g, num_bits, dtype) for g in fp]


def fp_load(compressed_fp, num_bits, dtype=torch.float32):
    """inverse of fp_savable (fp_def.py:258-263)"""
    return [load4fp(g, num_bits, dtype) for g in compressed_fp]


def fp_freeze(fp):
    """fp_def.py:266-268"""
    for g in fp:
        g.requires_grad = False
