"""Quantisers and the uint8 grid codec - the call surface of the reference's Projects/models.py, executed by
the HIP kernels nic_quantize / nic_quantize_to_bit / nic_clamp / nic_save4fp_u8 / nic_load4fp_u8.

Tensors must live on a HIP device (fp32); numpy inputs of the host-side helpers are accepted where the
reference accepts them (they are host post-processing of decoded images, image_compression.py:406-407).
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib


def _q_range(num_bits):
    return -(pow(2, num_bits) - 1) / pow(2, num_bits + 1), 1 / 2        # models.py:49-50


def _run(fn_name, src: torch.Tensor, num_bits: int, out_dtype=torch.float32):
    s = _lib.require_cuda_f32(src.detach(), "tensor")
    dst = torch.empty(s.shape, dtype=out_dtype, device=s.device)
    with torch.cuda.device(s.device):
        _lib.check(getattr(_lib.load(), fn_name)(_lib.ptr(s), _lib.ptr(dst), s.numel(), int(num_bits), _lib.stream_ptr(s.device)), fn_name)
    return dst


def scale_to_bit(tensor, bit=8):
    """normalised -> integer scale (models.py:5-7); host arithmetic on whatever array type comes in"""
    return tensor * (pow(2, bit) - 1)


def normalize_from_bit(tensor, bit=8):
    """models.py:11-13"""
    return tensor / (pow(2, bit) - 1)


def quantize(array, bit):
    """floor(x * (2^b - 1) + 1/2) / (2^b - 1)   (models.py:29-35)"""
    if isinstance(array, np.ndarray):
        return np.floor(array * (pow(2, bit) - 1) + 0.5) / (pow(2, bit) - 1)
    return _run("nic_quantize", array, bit)


def quantize_to_bit(array, num_bits=8):
    """quantize, then back to the integer scale 0..2^b-1 (models.py:39-40)"""
    if isinstance(array, np.ndarray):
        return scale_to_bit(quantize(array, num_bits), num_bits)
    return _run("nic_quantize_to_bit", array, num_bits)


def quantize_from_bit_to_bit(array, bit):
    """models.py:44-45 (host data preparation of 3D inputs, image_compression.py:449)"""
    return scale_to_bit(quantize(normalize_from_bit(array, bit), bit), bit)


def quantize_clamp(tensor, num_bits=8):
    """clamp to [q_min, 1/2] (models.py:48-51); returns a new tensor"""
    lo, hi = _q_range(num_bits)
    out = _lib.require_cuda_f32(tensor.detach(), "tensor").clone()
    with torch.cuda.device(out.device):
        _lib.check(_lib.load().nic_clamp(_lib.ptr(out), out.numel(), lo, hi, _lib.stream_ptr(out.device)), "nic_clamp")
    return out


def quantize4fp(tensor, num_bits):
    """grid quantiser (models.py:55-57)"""
    return _run("nic_quantize", tensor, num_bits)


def save4fp(tensor, num_bits, dtype=torch.uint8):
    """u = floor(x (2^b-1) + 1/2) + 2^(b-1) - 1 as uint8 (models.py:61-64)"""
    if dtype != torch.uint8:
        raise NotImplementedError("the codec stores uint8 (bits2dtype_torch(FP_BITS <= 8), image_compression.py:380)")
    return _run("nic_save4fp_u8", tensor, num_bits, torch.uint8)


def load4fp(tensor, num_bits, dtype=torch.float32):
    """(u - 2^(b-1) + 1) / (2^b-1)   (models.py:68-71)"""
    if tensor.dtype != torch.uint8 or not tensor.is_cuda:
        raise RuntimeError("load4fp expects a uint8 tensor on a HIP device")
    if dtype != torch.float32:
        raise NotImplementedError("fp32 only")
    t = tensor.contiguous()
    dst = torch.empty(t.shape, dtype=torch.float32, device=t.device)
    with torch.cuda.device(t.device):
        _lib.check(_lib.load().nic_load4fp_u8(_lib.ptr(t), _lib.ptr(dst), t.numel(), int(num_bits), _lib.stream_ptr(t.device)), "nic_load4fp_u8")
    return dst
