"""ctypes binding of libnicv2_hip.so (C ABI: include/nicv2_hip.h).

There is deliberately NO fallback: if the library is missing or a call fails, the caller gets a
RuntimeError.  Nothing in this package computes the hot path on the CPU or through eager torch ops.
"""
from __future__ import annotations

import ctypes
import os
import threading
from typing import Optional

import torch  # imported before the library so libamdhip64.so.7 resolves to the runtime torch already loaded

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NIC_LIB_PATH") or os.path.join(_HERE, "libnicv2_hip.so")   # override: A/B timing of kernel variants only

NIC_ABI_VERSION = 9
NIC_PE_TRIANGULAR, NIC_PE_SINUSOIDAL = 0, 1
NIC_G1_REFERENCE, NIC_G1_TEXTBOOK, NIC_G1_UNWEIGHTED = 0, 1, 2
NIC_NOISE_NONE, NIC_NOISE_TENSOR, NIC_NOISE_KERNEL = 0, 1, 2
NIC_FLAG_ORIGINS_ALIGNED = 1
NIC_FLAG_SPLIT_BF16 = 2
NIC_FLAG_SPLIT_TILE32 = 4
NIC_FLAG_MLPN = 8
NIC_FLAG_GRID_BF16 = 16
NIC_FLAG_GRID_FP16 = 32
NIC_FLAG_BF16 = 64
NIC_FLAG_FP16 = 128
NIC_FLAG_ORIGINS_HOST = 256          # fused entry points: `origins` is host memory (<= NIC_ORIGINS_INLINE_MAX crops), passed to the kernel by value
NIC_ORIGINS_INLINE_MAX = 16
NIC_MAX_LINEAR = 5


class NicPathDesc(ctypes.Structure):
    """struct nic_path_desc (include/nicv2_hip.h)."""
    _fields_ = [
        ("dim", ctypes.c_int32), ("method", ctypes.c_int32), ("channels", ctypes.c_int32),
        ("pe_channels", ctypes.c_int32), ("hidden", ctypes.c_int32), ("pe_mode", ctypes.c_int32),
        ("g1_weight_mode", ctypes.c_int32), ("log2_step", ctypes.c_int32), ("lod_value", ctypes.c_float),
        ("num_crops", ctypes.c_int32), ("extent", ctypes.c_int32 * 3), ("g0_nodes", ctypes.c_int32 * 3),
        ("g1_nodes", ctypes.c_int32 * 3), ("pe_div", ctypes.c_float * 8), ("noise_mode", ctypes.c_int32),
        ("num_bits", ctypes.c_int32), ("noise_seed", ctypes.c_uint64), ("noise_offset", ctypes.c_uint64),
        ("sample_base", ctypes.c_int64), ("loss_scale", ctypes.c_float), ("flags", ctypes.c_int32),
        ("passes", ctypes.c_int32), ("dz_scale_log2", ctypes.c_int32), ("max_workgroups", ctypes.c_int32),
        ("tail", ctypes.c_void_p),               # const nic_step_tail *: the optimiser step riding on the reduce launch (training entry points)
    ]


class NicMlp(ctypes.Structure):
    """struct nic_mlp: layer i in w[i] / b[i], n_linear = 3 (the reference's decoder) or 5"""
    _fields_ = [("w", ctypes.c_void_p * 5), ("b", ctypes.c_void_p * 5), ("n_linear", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class NicMlpGrads(ctypes.Structure):
    _fields_ = [("w", ctypes.c_void_p * 5), ("b", ctypes.c_void_p * 5)]


NIC_ML_MAX_LEVELS = 5


class NicMlPairs(ctypes.Structure):
    """struct nic_ml_pairs (include/nicv2_hip.h): the level pairs of a multi-level launch"""
    _fields_ = [("levels", ctypes.c_int32), ("reserved", ctypes.c_int32), ("g0", ctypes.c_void_p * NIC_ML_MAX_LEVELS), ("g1", ctypes.c_void_p * NIC_ML_MAX_LEVELS),
                ("g0_grad", ctypes.c_void_p * NIC_ML_MAX_LEVELS), ("g1_grad", ctypes.c_void_p * NIC_ML_MAX_LEVELS),
                ("g0_nodes", (ctypes.c_int32 * 2) * NIC_ML_MAX_LEVELS), ("g1_nodes", (ctypes.c_int32 * 2) * NIC_ML_MAX_LEVELS)]


NIC_ADAM_MAX_TENSORS = 32
NIC_ADAM_ZERO_GRAD = 1
NIC_ADAM_SCHED_COL1 = 2


class NicAdamTensor(ctypes.Structure):
    """struct nic_adam_tensor (include/nicv2_hip.h)."""
    _fields_ = [("param", ctypes.c_void_p), ("grad", ctypes.c_void_p), ("exp_avg", ctypes.c_void_p), ("exp_avg_sq", ctypes.c_void_p),
                ("n", ctypes.c_int64), ("step", ctypes.c_int64), ("lr", ctypes.c_double), ("clamp_lo", ctypes.c_float),
                ("clamp_hi", ctypes.c_float), ("param16", ctypes.c_void_p), ("param16_kind", ctypes.c_int32), ("flags", ctypes.c_int32),
                ("reps", ctypes.c_int32), ("reserved", ctypes.c_int32), ("rep_stride", ctypes.c_int64), ("state_rep_stride", ctypes.c_int64)]


class NicStepTail(ctypes.Structure):
    """struct nic_step_tail (include/nicv2_hip.h): the optimiser step as the tail of a fused training step"""
    _fields_ = [("tensors", ctypes.c_void_p), ("count", ctypes.c_int32), ("n_stream", ctypes.c_int32), ("beta1", ctypes.c_double),
                ("beta2", ctypes.c_double), ("eps", ctypes.c_double), ("sched", ctypes.c_void_p), ("sched_rows", ctypes.c_int64)]


NIC_STRIPE_MAX_ROWS = 15


class NicRowSet(ctypes.Structure):
    """struct nic_row_set (include/nicv2_hip.h)."""
    _fields_ = [("base", ctypes.c_void_p), ("plane", ctypes.c_int64), ("row_elems", ctypes.c_int32), ("channels", ctypes.c_int32),
                ("nrows", ctypes.c_int32), ("rows", ctypes.c_int32 * NIC_STRIPE_MAX_ROWS)]


class NicTargetImage(ctypes.Structure):
    """struct nic_target_image (include/nicv2_hip.h)."""
    _fields_ = [("data", ctypes.c_void_p), ("is_u8", ctypes.c_int32), ("den", ctypes.c_float), ("size", ctypes.c_int32 * 3),
                ("reserved", ctypes.c_int32)]


_P, _I, _L, _F, _SZ, _DBL = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_size_t, ctypes.c_double
_D = ctypes.POINTER(NicPathDesc)
_M = ctypes.POINTER(NicMlp)
_G = ctypes.POINTER(NicMlpGrads)

# name -> (restype, argtypes): every symbol include/nicv2_hip.h declares
SIGNATURES = {
    "nic_abi_version": (_I, []),
    "nic_error_string": (ctypes.c_char_p, [_I]),
    "nic_decoder_input_channels": (_I, [_I, _I, _I, _I]),
    "nic_workspace_bytes": (_SZ, [_D]),
    "nic_encode": (_I, [_D, _P, _P, _P, _P, _P]),
    "nic_encode_split": (_I, [_D, _P, _P, _P, _P, _P]),
    "nic_encode_backward": (_I, [_D, _P, _P, _P, _P, _P]),
    "nic_positional_encoding": (_I, [_P, _L, _I, _I, _I, ctypes.POINTER(ctypes.c_float), _P, _P]),
    "nic_lut_gather": (_I, [_P, _I, _I, _P, _L, _L, _P, _P]),
    "nic_decoder_forward": (_I, [_M, _P, _L, _I, _I, _P, _P]),
    "nic_decoder_backward": (_I, [_M, _P, _P, _L, _I, _I, _P, _G, _P, _SZ, _P]),
    "nic_decoder_general_workspace_bytes": (_SZ, [_L, _I, _I, _I, _I]),
    "nic_decoder_general_forward": (_I, [_M, _P, _L, _I, _I, _P, _P, _SZ, _P]),
    "nic_decoder_general_backward": (_I, [_M, _P, _P, _L, _I, _I, _P, _G, _P, _SZ, _P]),
    "nic_decoder_pad": (_I, [_M, _I, _I, _I, _G, _P]),
    "nic_decoder_unpad": (_I, [_G, _I, _I, _I, _I, _G, _P]),
    "nic_fused_forward": (_I, [_D, _P, _P, _P, _M, _P, _P, _P]),
    "nic_fused_forward_u8": (_I, [_D, _P, _P, _P, _M, _P, _P, _P]),
    "nic_fused_forward_backward": (_I, [_D, _P, _P, _P, _M, _P, _P, _P, _P, _P, _P, _G, _P, _SZ, _P]),
    "nic_fused_forward_backward_img": (_I, [_D, _P, _P, _P, _M, _P, ctypes.POINTER(NicTargetImage), _P, _P, _P, _P, _G, _P, _SZ, _P]),
    "nic_fused_backward_dy": (_I, [_D, _P, _P, _P, _M, _P, _P, _P, _P, _G, _P, _SZ, _P]),
    "nic_quantize": (_I, [_P, _P, _L, _I, _P]),
    "nic_quantize_to_bit": (_I, [_P, _P, _L, _I, _P]),
    "nic_clamp": (_I, [_P, _L, _F, _F, _P]),
    "nic_save4fp_u8": (_I, [_P, _P, _L, _I, _P]),
    "nic_load4fp_u8": (_I, [_P, _P, _L, _I, _P]),
    "nic_psnr": (_I, [_P, _P, _L, _I, _P, _P, _SZ, _P]),
    "nic_adam_step": (_I, [_P, _P, _P, _P, _L, _DBL, _DBL, _DBL, _DBL, _L, _F, _F, _P]),
    "nic_gather_corners": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _L, _I, _P, _P]),
    "nic_adam_multi": (_I, [ctypes.POINTER(NicAdamTensor), _I, _DBL, _DBL, _DBL, _P]),
    "nic_sampler_step_begin": (_I, [ctypes.c_uint64, _P, _I, _I, ctypes.c_int32, _P, _P, _P, _L, _P]),
    "nic_fused_forward_backward_img_dev": (_I, [_D, _P, _P, _P, _M, ctypes.POINTER(NicTargetImage), _P, _P, _P, _G, _P, _P, _SZ, _P]),
    "nic_adam_multi_dev": (_I, [ctypes.POINTER(NicAdamTensor), _I, _DBL, _DBL, _DBL, _P, _L, _P, _P]),
    "nic_sampler_lod_host": (_I, [ctypes.c_uint64, ctypes.c_uint64, _I, _I]),
    "nic_sampler_origins_host": (_I, [ctypes.c_uint64, ctypes.c_uint64, _I, _I, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32)]),
    "nic_sampler_draw_origins": (_I, [ctypes.c_uint64, ctypes.c_uint64, _I, _I, ctypes.c_int32, _P, _P]),
    "nic_rgbx_interleave": (_I, [_P, _L, _P, _P]),
    "nic_rgbx_downsample2": (_I, [_P, _I, _I, _P, _P]),
    "nic_rgbx_resample_axis": (_I, [_P, _I, _I, _I, _I, _P, _P, _I, _P, _P]),
    "nic_fused_ml_forward_backward": (_I, [_D, ctypes.POINTER(NicMlPairs), _P, _M, _P, _P, _P, _P, _G, _P, _SZ, _P]),
    "nic_fused_ml_forward": (_I, [_D, ctypes.POINTER(NicMlPairs), _P, _M, _P, _P]),
    "nic_mark_kernel_end": (_I, [_P]),
    "nic_stripe_pack": (_I, [_P, _L, ctypes.POINTER(NicRowSet), _I, _P, _P]),
    "nic_stripe_unpack": (_I, [_P, _L, ctypes.POINTER(NicRowSet), _I, _P, _P]),
}

_lib: Optional[ctypes.CDLL] = None
_lock = threading.Lock()


def load() -> ctypes.CDLL:
    """Loads the library (once).  Raises if it has not been built: there is no other implementation."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise RuntimeError(
                    f"{LIB_PATH} is missing - the HIP extension is the only implementation of this path. "
                    "Build it with `python -m neural_image_compression_v2_amd._build` (hipcc, gfx950).")
            lib = ctypes.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(lib, name)          # AttributeError if the build is stale: loud by design
                fn.restype = res
                fn.argtypes = args
            if lib.nic_abi_version() != NIC_ABI_VERSION:
                raise RuntimeError("libnicv2_hip.so ABI version mismatch - rebuild")
            _lib = lib
    return _lib


class Unsupported(RuntimeError):
    """NIC_E_UNSUPPORTED: no specialised kernel for this dim / method / channels / hidden / depth combination (nothing was launched).
    The host loop answers it with the layer-wise general route (nic_encode + nic_decoder_general_*)."""


NIC_E_UNSUPPORTED = -2


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().nic_error_string(int(rc)).decode()
        raise (Unsupported if rc == NIC_E_UNSUPPORTED else RuntimeError)(f"libnicv2_hip {what} failed: {msg} (code {rc})")


def stream_ptr(device: Optional[torch.device] = None) -> ctypes.c_void_p:
    """the current torch HIP stream as the void* the C ABI takes"""
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_cuda_f32(t: torch.Tensor, name: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name} lives on {t.device}: this package only runs on a HIP device (no CPU path)")
    if t.dtype != torch.float32:
        raise NotImplementedError(f"{name} has dtype {t.dtype}; the gfx950 kernels are fp32 "
                                  "(the reference's 16-bit path is itself unfinished, readme.md:9)")
    return t if t.is_contiguous() else t.contiguous()


GRID_DTYPES = (torch.float32, torch.bfloat16, torch.float16)


def require_cuda_grid(t: torch.Tensor, name: str) -> torch.Tensor:
    """grids: fp32, or 16-bit STORAGE (bfloat16 / float16; NIC_FLAG_GRID_*) - the arithmetic is fp32 either way"""
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name} lives on {t.device}: this package only runs on a HIP device (no CPU path)")
    if t.dtype not in GRID_DTYPES:
        raise NotImplementedError(f"{name} has dtype {t.dtype}: grids are fp32, or bfloat16 / float16 storage")
    return t if t.is_contiguous() else t.contiguous()


def ptr(t: Optional[torch.Tensor]) -> ctypes.c_void_p:
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


_workspaces = {}


def workspace(device: torch.device, nbytes: int) -> torch.Tensor:
    """per (device, stream) scratch buffer, grown on demand; owned by torch's caching allocator"""
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(device).cuda_stream)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf
