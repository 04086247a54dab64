"""Positional encodings, PSNR and dtype helpers - the hot-path part of the reference's Projects/utils.py
(utils.py:117-130, 198-284, 301-328), executed by nic_positional_encoding / nic_lut_gather / nic_psnr.
The reference file's host I/O helpers (video reading, csv dumps, log tee) are outside this path.
"""
from __future__ import annotations

import ctypes
import math

import numpy as np
import torch

from . import _lib
from .fused import sinusoidal_div_term


def _pe(coord, num_channels, mode):
    if isinstance(coord, (tuple, list)):
        coord = torch.stack([torch.as_tensor(c) for c in coord])
    c = _lib.require_cuda_f32(coord.to(torch.float32) if coord.dtype != torch.float32 else coord, "coord")
    D, n = c.shape
    out = torch.empty(num_channels * D, n, dtype=torch.float32, device=c.device)
    div = (ctypes.c_float * 8)(*(sinusoidal_div_term(num_channels) + [0.0] * 8)[:8])
    with torch.cuda.device(c.device):
        _lib.check(_lib.load().nic_positional_encoding(_lib.ptr(c), n, D, int(num_channels), mode, div, _lib.ptr(out),
                                                       _lib.stream_ptr(c.device)), "nic_positional_encoding")
    return out


def positional_encoding(coord, num_channels, device=None, dtype=None):
    """sinusoidal PE: tuple of D coordinate vectors [n] -> [P*D, n] (utils.py:198-208)"""
    return _pe(coord, num_channels, _lib.NIC_PE_SINUSOIDAL)


def triangular_positional_encoding(coord, num_channels, device=None, dtype=None):
    """triangular-wave PE: [D, n] -> [P*D, n] (utils.py:211-223)"""
    return _pe(coord, num_channels, _lib.NIC_PE_TRIANGULAR)


def tri(x, offset=0.5):
    """2|((x - o) mod 2) - 1| - 1 (utils.py:226-227) = row P-2 of a one-dimensional triangular PE"""
    flat = x.reshape(1, -1).to(torch.float32) - float(offset)
    return _pe(flat, 2, _lib.NIC_PE_TRIANGULAR)[0].reshape(x.shape)


def triangular_positional_encoding_1d(device, dtype=torch.float32, sequence_length=8, octaves=3, include_constant=True):
    """the LUT [2*octaves-1 (+1), L] (utils.py:230-243): rows tri(x,0), tri(x/2,0), tri(x/2,.5), ..."""
    x = torch.arange(0, sequence_length, dtype=torch.float32, device=device)
    rows = []
    for octave in range(octaves):
        for i, off in enumerate((0.0, 0.5)):
            if octave == 0 and i == 1:
                continue
            rows.append(tri(x / (2 ** octave), off))
    if include_constant:
        rows.append(torch.zeros(sequence_length, dtype=torch.float32, device=device))
    return torch.stack(rows)


def lut_gather(encodings, coordinates):
    """encodings[:, coord % L] -> [b, rows, L'] (positional_encoding.py:36-42)"""
    lut = _lib.require_cuda_f32(encodings, "encodings")
    c = coordinates.to(torch.int64).contiguous()
    if not c.is_cuda:
        raise RuntimeError("coordinates must be on the HIP device")
    b, L = c.shape
    out = torch.empty(b, lut.shape[0], L, dtype=torch.float32, device=lut.device)
    with torch.cuda.device(lut.device):
        _lib.check(_lib.load().nic_lut_gather(_lib.ptr(lut), lut.shape[0], lut.shape[1], _lib.ptr(c), b, L, _lib.ptr(out),
                                              _lib.stream_ptr(lut.device)), "nic_lut_gather")
    return out


def convert_coordinate_start(coordinate_start, h, w, device=None, dtype=None, stride=1, flatten_sequence=True):
    """[b, 2] tile origins -> per-sample (x, y) integer coordinates, x outermost (utils.py:266-284; h == w like the
    reference, whose .view() of a meshgrid only admits the square case).  Index bookkeeping only."""
    if h != w:
        raise ValueError("convert_coordinate_start: the reference only supports h == w (utils.py:268-270)")
    dev = coordinate_start.device
    xo = torch.arange(0, w * stride, stride, device=dev)
    yo = torch.arange(0, h * stride, stride, device=dev)
    b = coordinate_start.shape[0]
    fx = coordinate_start[:, 0].reshape(b, 1, 1) + xo.reshape(1, w, 1).expand(1, w, h)
    fy = coordinate_start[:, 1].reshape(b, 1, 1) + yo.reshape(1, 1, h).expand(1, w, h)
    if flatten_sequence:
        return fx.reshape(b, -1), fy.reshape(b, -1)
    return fx.reshape(b, h, w, 1), fy.reshape(b, h, w, 1)


def triangular_positional_encoding_2d(coordinates, h, w, device=None, dtype=torch.float32, sequence_length=8, octaves=3, stride=1,
                                      include_constant=True):
    """[b, 2] -> [b, 2*rows, h, w], x block then y block (utils.py:246-263)"""
    lut = triangular_positional_encoding_1d(coordinates.device, torch.float32, sequence_length, octaves, include_constant)
    fx, fy = convert_coordinate_start(coordinates, h, w, stride=stride)
    b = coordinates.shape[0]
    return torch.cat([lut_gather(lut, fx).view(b, -1, h, w), lut_gather(lut, fy).view(b, -1, h, w)], dim=1)


def calculate_psnr(original, reconstructed, num_bits=8):
    """10 log10(peak^2 / mse), peak = 2^num_bits (NOT 2^b - 1)  (utils.py:117-130).  Device tensors are reduced on the
    GPU and a 0-dim device tensor is returned (no host sync); numpy inputs follow the reference's numpy branch."""
    if isinstance(original, np.ndarray):
        mse = np.mean((original - reconstructed) ** 2)
        if mse == 0:
            return float("inf")
        return 10 * np.log10(pow(2, num_bits) ** 2 / mse)
    a = _lib.require_cuda_f32(original.detach(), "original")
    b = _lib.require_cuda_f32(reconstructed.detach(), "reconstructed")
    if a.shape != b.shape:
        raise ValueError("shape mismatch")
    out = torch.empty(2, dtype=torch.float32, device=a.device)
    ws = _lib.workspace(a.device, 1024 * 8)
    with torch.cuda.device(a.device):
        _lib.check(_lib.load().nic_psnr(_lib.ptr(a), _lib.ptr(b), a.numel(), int(num_bits), _lib.ptr(out), _lib.ptr(ws), ws.numel(),
                                        _lib.stream_ptr(a.device)), "nic_psnr")
    return out[1]


def bits2dtype_torch(num_bits, dtype="float"):
    """utils.py:301-313"""
    if num_bits <= 8:
        return torch.uint8
    if num_bits == 16:
        return {"int": torch.int16, "uint": torch.uint16, "float": torch.float16}[dtype]
    if num_bits == 32:
        return torch.float32
    if num_bits == 64:
        return torch.float64
    return None


def bits2dtype_np(num_bits, dtype="float"):
    """utils.py:316-328"""
    if num_bits <= 8:
        return np.uint8
    if num_bits == 16:
        return {"int": np.int16, "uint": np.uint16, "float": np.float16}[dtype]
    if num_bits == 32:
        return np.float32
    if num_bits == 64:
        return np.float64
    return None


def judge_value(arg, dtype, error_massage=""):
    """``NAME=value`` parsing of the launch scripts (utils.py:13-31), returned as a Python value (no exec)"""
    v = arg.split("=", 1)[1]
    if dtype == "int":
        return int(v)
    if dtype == "float":
        return float(v)
    if dtype == "bool":
        if v.lower() in ("true", "1"):
            return True
        if v.lower() in ("false", "0"):
            return False
        raise ValueError(f"{error_massage} must be a boolean (True/False or 1/0)")
    return v
