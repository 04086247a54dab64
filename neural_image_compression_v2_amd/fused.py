"""Host side of the fused gfx950 kernels: geometry descriptors, argument validation, autograd glue.

Everything here is plumbing above the C ABI (include/nicv2_hip.h); the arithmetic of the path runs in
libnicv2_hip.so.  Reference citations are relative to /root/reference/Projects.
"""
from __future__ import annotations

import ctypes
import dataclasses
import functools
import math
import os
from dataclasses import dataclass, field, replace
from typing import List, Optional, Sequence, Tuple, Union

import torch

from . import _lib
from ._lib import (NIC_G1_REFERENCE, NIC_G1_TEXTBOOK, NIC_G1_UNWEIGHTED, NIC_NOISE_NONE, NIC_NOISE_KERNEL,
                   NIC_NOISE_TENSOR, NIC_PE_SINUSOIDAL, NIC_PE_TRIANGULAR)


def _on_tensor_device(fn):
    """Runs ``fn`` with the device of its first HIP tensor argument current: the kernels launch on that device's stream, and a
    launch on a stream of a device that is not the current one fails (multi-GPU processes, config 5's per-GPU fits)."""
    @functools.wraps(fn)
    def wrapper(*args, **kw):
        for a in args:
            if isinstance(a, torch.Tensor) and a.is_cuda:
                with torch.cuda.device(a.device):
                    return fn(*args, **kw)
        return fn(*args, **kw)
    return wrapper


def g1_weights_enabled(step_number) -> bool:
    """the reference's guard, evaluated the way it writes it (fp_def.py:136,170,209)"""
    return int(1 // (step_number / 2)) != 1


def log2_step_of(step_number) -> int:
    e = round(math.log2(step_number))
    if 2.0 ** e != float(step_number):
        raise ValueError(f"step_number must be a power of two (image_compression.py:79), got {step_number}")
    return int(e)


def sinusoidal_div_term(num_channels: int) -> List[float]:
    """div_term exactly as torch evaluates it in fp32 (utils.py:202); 3 numbers, computed on the host"""
    t = torch.exp(torch.arange(0, num_channels, 2, dtype=torch.float32) * -(math.log(10000.0) / num_channels))
    return [float(v) for v in t]


def decoder_input_channels(dim: int, method: int, channels: int, pe_channels: int) -> int:
    """Cin (var2.py:114-118)"""
    k0 = 4 if (dim == 2 or method == 4) else 8
    return channels * (k0 + 1) + pe_channels * dim + 1


@dataclass
class PathGeometry:
    """Python mirror of nic_path_desc: which grid pair, which samples (arguments of create_g0_g1* and
    create_decoder_input_*, fp_def.py:115-223, image_compression.py:71-167)."""
    dim: int
    method: int                      # 1: 2D, 3: 3D full corners, 4: 3D tetrahedral G0
    step_number: Union[int, float]   # 2^(mip - 2(fl+1))
    mip_level: float
    extent: Tuple[int, ...]          # samples per axis of a crop (x first)
    num_crops: int
    channels: int = 12
    pe_channels: int = 6
    hidden: int = 64
    use_tri_pe: bool = True
    textbook_weights: bool = False
    noise_mode: int = NIC_NOISE_NONE
    num_bits: int = 8
    noise_seed: int = 0
    noise_offset: int = 0
    sample_base: int = 0
    loss_scale: Optional[float] = None   # default 1 / (3 N)
    flags: int = 0                       # NIC_FLAG_* the caller vouches for (the wrappers add ORIGINS_ALIGNED when they can see it)
    split_bf16: bool = False             # 2D training steps: matrix products as hi + lo bf16 pairs on the bf16 pipe (NIC_FLAG_SPLIT_BF16)
    passes: int = 1                      # training steps: every crop sampled `passes` times in one launch (nic_path_desc.passes)
    split_tile32: bool = False           # with split_bf16, 2D training: the 4-wave x 32-sample kernel instead of the 8-wave x 16-sample default
    max_workgroups: int = 0              # > 0: the launch takes at most this many workgroups (concurrent fits on separate streams share the CUs)
    mlpn: bool = False                   # 3-layer decoders on the depth-generic kernel that serves 5-layer ones (NIC_FLAG_MLPN: cross-check)
    bf16: bool = False                   # training steps: PLAIN bf16 products (NIC_FLAG_BF16; every layout, 3 or 5 Linear layers) - the north star's "bf16"
    fp16: bool = False                   # .. on IEEE half operands instead (NIC_FLAG_FP16: the reference's own 16-bit type, utils.py:301-313); default channel counts
    dz_scale_log2: int = 0               # fp16: dZ is carried as 2^k dZ; 0 = chosen from loss_scale by the MSE entry points (nic_path_desc.dz_scale_log2)

    def __post_init__(self):
        if self.dim == 3 and self.method == 3:
            self.use_tri_pe = True           # fp_def.py:169
        if self.dim == 3 and self.method == 4:
            self.use_tri_pe = False          # fp_def.py:208
        if len(self.extent) != self.dim:
            raise ValueError("extent needs one entry per axis")
        if int(self.passes) < 1:
            raise ValueError("passes must be >= 1")

    @property
    def n_per_crop(self) -> int:
        n = 1
        for e in self.extent:
            n *= int(e)
        return n

    @property
    def n_samples(self) -> int:
        return self.n_per_crop * self.num_crops * int(self.passes)

    @property
    def cin(self) -> int:
        return decoder_input_channels(self.dim, self.method, self.channels, self.pe_channels)

    def g1_mode(self) -> int:
        if not g1_weights_enabled(self.step_number):
            return NIC_G1_UNWEIGHTED                                   # Q6
        return NIC_G1_TEXTBOOK if self.textbook_weights else NIC_G1_REFERENCE

    def to_desc(self, g0: torch.Tensor, g1: torch.Tensor, aligned: bool = False) -> _lib.NicPathDesc:
        d = _lib.NicPathDesc()
        d.dim, d.method = self.dim, self.method
        d.channels, d.pe_channels, d.hidden = self.channels, self.pe_channels, self.hidden
        d.pe_mode = NIC_PE_TRIANGULAR if self.use_tri_pe else NIC_PE_SINUSOIDAL
        d.g1_weight_mode = self.g1_mode()
        d.log2_step = log2_step_of(self.step_number)
        d.lod_value = float(self.mip_level)
        d.num_crops = int(self.num_crops)
        for a in range(3):
            d.extent[a] = int(self.extent[a]) if a < self.dim else 1
            # tensor is [C, (Z,) Y, X]: axis a of the samples is tensor dim -(a+1)
            d.g0_nodes[a] = int(g0.shape[-(a + 1)]) if a < self.dim else 1
            d.g1_nodes[a] = int(g1.shape[-(a + 1)]) if a < self.dim else 1
        div = sinusoidal_div_term(self.pe_channels)
        for i in range(8):
            d.pe_div[i] = div[i] if i < len(div) else 0.0
        d.noise_mode = int(self.noise_mode)
        d.num_bits = int(self.num_bits)
        d.noise_seed = int(self.noise_seed) & 0xFFFFFFFFFFFFFFFF
        d.noise_offset = int(self.noise_offset) & 0xFFFFFFFFFFFFFFFF
        d.sample_base = int(self.sample_base)
        d.loss_scale = float(self.loss_scale) if self.loss_scale is not None else 1.0 / (3.0 * self.n_samples)
        d.flags = int(self.flags) | (_lib.NIC_FLAG_ORIGINS_ALIGNED if aligned else 0)
        d.passes = int(self.passes)
        d.max_workgroups = int(self.max_workgroups)
        if self.split_bf16 or os.environ.get("NIC_FORCE_SPLIT_BF16") == "1":   # env: test switch for the whole suite
            d.flags |= _lib.NIC_FLAG_SPLIT_BF16
        if self.split_tile32:
            d.flags |= _lib.NIC_FLAG_SPLIT_TILE32
        if self.mlpn:
            d.flags |= _lib.NIC_FLAG_MLPN
        if self.bf16:
            d.flags |= _lib.NIC_FLAG_BF16
        if self.fp16:
            d.flags |= _lib.NIC_FLAG_FP16
        d.dz_scale_log2 = int(self.dz_scale_log2)
        if g0.dtype != g1.dtype:
            raise ValueError("G0 and G1 must share a dtype")
        if g0.dtype == torch.bfloat16:
            d.flags |= _lib.NIC_FLAG_GRID_BF16                            # 16-bit grid STORAGE: widened in the gather, fp32 arithmetic and gradients
        elif g0.dtype == torch.float16:
            d.flags |= _lib.NIC_FLAG_GRID_FP16
        return d


def grid_storage(g: torch.Tensor) -> torch.Tensor:
    """the tensor the kernels gather from: the 16-bit mirror of an fp32 master created by ``create_pyramid(dtype=float16 | bfloat16)``
    (``master.mirror16``), otherwise the tensor itself"""
    m = getattr(g, "mirror16", None)
    return g if m is None else m


def check_grids(geo: PathGeometry, g0: torch.Tensor, g1: torch.Tensor) -> None:
    for name, g in (("G0", g0), ("G1", g1)):
        if g.dim() != geo.dim + 1 or g.shape[0] != geo.channels:
            raise ValueError(f"{name} must be [C={geo.channels}" + ", n" * geo.dim + f"], got {tuple(g.shape)}")


def check_origins(geo: PathGeometry, origins_host: torch.Tensor, g0: torch.Tensor, g1: torch.Tensor) -> None:
    """Host-side bounds check: every corner index must stay inside its grid (the reference would raise
    IndexError from advanced indexing; the kernels clamp for memory safety only)."""
    o = origins_host.reshape(-1, geo.dim).to(torch.int64)
    if o.shape[0] != geo.num_crops:
        raise ValueError(f"need {geo.num_crops} origins, got {o.shape[0]}")
    if bool((o < 0).any()):
        raise IndexError("negative crop origin")
    s = float(geo.step_number)
    for a in range(geo.dim):
        qmax = int(o[:, a].max()) + int(geo.extent[a]) - 1
        i0 = math.floor(qmax * s) + 1
        i1 = math.floor(qmax * s / 2) + 1
        n0, n1 = int(g0.shape[-(a + 1)]), int(g1.shape[-(a + 1)])
        if i0 > n0 - 1 or i1 > n1 - 1:
            raise IndexError(f"axis {a}: corner index {i0}/{i1} outside grids with {n0}/{n1} nodes "
                             f"(origin+extent {qmax + 1}, step {geo.step_number})")


def upload_origins(geo: PathGeometry, coord, device, g0=None, g1=None) -> torch.Tensor:
    """int32 [num_crops, dim] on the device.  Host inputs (list / CPU tensor) are validated first; a
    device tensor is taken as is (validating it would force a sync)."""
    if isinstance(coord, torch.Tensor) and coord.is_cuda:
        return coord.reshape(-1, geo.dim).to(torch.int32).contiguous()
    host = torch.as_tensor(coord).reshape(-1, geo.dim).to(torch.int64)
    if g0 is not None:
        check_origins(geo, host, g0, g1)
    return host.to(torch.int32).to(device, non_blocking=True)


def origins_aligned(geo: PathGeometry, coord) -> bool:
    """True when host-visible origins are all multiples of the cell size 1 / step_number (whole-image passes, decode tiles): the
    launch then needs no extra cell block per axis (NIC_FLAG_ORIGINS_ALIGNED).  Device-resident origins are not inspected."""
    if isinstance(coord, torch.Tensor) and coord.is_cuda:
        return False
    cell = int(round(1.0 / geo.step_number)) if geo.step_number < 1 else 1
    return bool((torch.as_tensor(coord).reshape(-1, geo.dim).to(torch.int64) % cell == 0).all())


def _mlp_struct(params: Sequence[torch.Tensor]) -> _lib.NicMlp:
    """params in nn.Sequential order: W1, b1, W2, b2, ... (3 or 5 Linear layers)"""
    m = _lib.NicMlp()
    nl = len(params) // 2
    for i in range(nl):
        m.w[i] = params[2 * i].data_ptr()
        m.b[i] = params[2 * i + 1].data_ptr()
    m.n_linear = nl
    return m


def _grads_struct(grads: Sequence[Optional[torch.Tensor]]) -> _lib.NicMlpGrads:
    g = _lib.NicMlpGrads()
    for i in range(len(grads) // 2):
        g.w[i] = 0 if grads[2 * i] is None else grads[2 * i].data_ptr()
        g.b[i] = 0 if grads[2 * i + 1] is None else grads[2 * i + 1].data_ptr()
    return g


def check_mlp(params: Sequence[torch.Tensor], cin: int, hidden: int) -> List[torch.Tensor]:
    params = list(params)
    if len(params) % 2 or not 2 <= len(params) // 2 <= _lib.NIC_MAX_LINEAR:
        raise NotImplementedError(f"decoders of 2 .. {_lib.NIC_MAX_LINEAR} Linear layers (the reference has 3, image_compression.py:57-64)")
    nl = len(params) // 2
    shapes = [(hidden, cin), (hidden,)] + [(hidden, hidden), (hidden,)] * (nl - 2) + [(3, hidden), (3,)]
    names = [f"{k}{i + 1}" for i in range(nl) for k in ("W", "b")]
    out = []
    for t, s, n in zip(params, shapes, names):
        t = _lib.require_cuda_f32(t, n)
        if tuple(t.shape) != s:
            raise ValueError(f"decoder parameter {n} has shape {tuple(t.shape)}, expected {s}")
        out.append(t)
    return out


FUSED_HIDDEN = 64          # the hidden width the fused kernels are built for


class PaddedMlp:
    """HIDDEN_LAYER_CHANNELS below 64 on the fused kernels: zero-padded copies of the decoder's parameters (``nic_decoder_pad``) are what the
    kernel reads, with ``desc.hidden = 64``; the real block of every padded gradient is copied back (``nic_decoder_unpad``).  Exactly the
    narrower decoder: the extra hidden units have zero weights and biases - pre-activation 0, GELU(0) = 0 - and zero outgoing weights, so they
    contribute nothing forward and receive exact zero gradients.  One small launch before and one after the fused step; buffers are persistent
    (``StepPlan`` keeps one), nothing is allocated per step."""

    def __init__(self, params: Sequence[torch.Tensor], cin: int, hidden: int, want_grads: bool):
        self.real = list(params)
        self.cin, self.hidden, self.nl = int(cin), int(hidden), len(params) // 2
        dev = params[0].device
        HP = FUSED_HIDDEN
        shapes = [(HP, cin), (HP,)] + [(HP, HP), (HP,)] * (self.nl - 2) + [(3, HP), (3,)]
        self.params = [torch.empty(sh, dtype=torch.float32, device=dev) for sh in shapes]
        self.grads = [torch.empty(sh, dtype=torch.float32, device=dev) for sh in shapes] if want_grads else None
        self.m_real = _mlp_struct(self.real)
        self.m = _mlp_struct(self.params)
        self.as_dst = _grads_struct(self.params)
        self.gs = _grads_struct(self.grads) if want_grads else None
        self.dev = dev

    @staticmethod
    def maybe(geo: "PathGeometry", params: Sequence[torch.Tensor], want_grads: bool) -> Optional["PaddedMlp"]:
        """None when the decoder already has the kernels' width (or is wider: the C side answers NIC_E_UNSUPPORTED and the caller falls back)"""
        return PaddedMlp(params, geo.cin, geo.hidden, want_grads) if 1 <= int(geo.hidden) < FUSED_HIDDEN else None

    def pad(self) -> None:
        _lib.check(_lib.load().nic_decoder_pad(ctypes.byref(self.m_real), self.cin, self.hidden, FUSED_HIDDEN, ctypes.byref(self.as_dst),
                                               _lib.stream_ptr(self.dev)), "nic_decoder_pad")

    def unpad(self, dst_grads: Sequence[Optional[torch.Tensor]]) -> None:
        gd = _grads_struct(dst_grads)
        _lib.check(_lib.load().nic_decoder_unpad(ctypes.byref(self.gs), self.nl, self.cin, self.hidden, FUSED_HIDDEN, ctypes.byref(gd),
                                                 _lib.stream_ptr(self.dev)), "nic_decoder_unpad")


# ------------------------------------------------------------------------------------------------------
# plain calls
# ------------------------------------------------------------------------------------------------------

@_on_tensor_device
def encode(geo: PathGeometry, g0: torch.Tensor, g1: torch.Tensor, coord) -> torch.Tensor:
    """[N, Cin] decoder input (create_decoder_input_* / finally_decode_input_*, image_compression.py:71-211)"""
    g0 = _lib.require_cuda_f32(g0.detach(), "G0")
    g1 = _lib.require_cuda_f32(g1.detach(), "G1")
    check_grids(geo, g0, g1)
    org = upload_origins(geo, coord, g0.device, g0, g1)
    out = torch.empty(geo.n_samples, geo.cin, dtype=torch.float32, device=g0.device)
    d = geo.to_desc(g0, g1)
    _lib.check(_lib.load().nic_encode(ctypes.byref(d), _lib.ptr(g0), _lib.ptr(g1), _lib.ptr(org), _lib.ptr(out),
                                      _lib.stream_ptr(g0.device)), "nic_encode")
    return out


@_on_tensor_device
def encode_split(geo: PathGeometry, g0: torch.Tensor, g1: torch.Tensor, coord) -> torch.Tensor:
    """[K0*C + K1*C + P*D, n] rows for ONE crop (create_g0_g1*, fp_def.py:115-223)"""
    g0 = _lib.require_cuda_f32(g0.detach(), "G0")
    g1 = _lib.require_cuda_f32(g1.detach(), "G1")
    check_grids(geo, g0, g1)
    org = upload_origins(geo, coord, g0.device, g0, g1)
    k0 = 4 if (geo.dim == 2 or geo.method == 4) else 8
    k1 = 4 if geo.dim == 2 else 8
    rows = (k0 + k1) * geo.channels + geo.pe_channels * geo.dim
    out = torch.empty(rows, geo.n_per_crop, dtype=torch.float32, device=g0.device)
    d = geo.to_desc(g0, g1)
    _lib.check(_lib.load().nic_encode_split(ctypes.byref(d), _lib.ptr(g0), _lib.ptr(g1), _lib.ptr(org), _lib.ptr(out),
                                            _lib.stream_ptr(g0.device)), "nic_encode_split")
    return out


@_on_tensor_device
def fused_forward(geo: PathGeometry, g0, g1, coord, params, noise: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[N, 3]: encode (+ noise) + decoder in one kernel (decode_image's inner step, image_compression.py:313-345)"""
    g0 = _lib.require_cuda_grid(grid_storage(g0).detach(), "G0")
    g1 = _lib.require_cuda_grid(grid_storage(g1).detach(), "G1")
    check_grids(geo, g0, g1)
    params = check_mlp([p.detach() for p in params], geo.cin, geo.hidden)
    org = upload_origins(geo, coord, g0.device, g0, g1)
    if geo.noise_mode == NIC_NOISE_TENSOR:
        noise = _lib.require_cuda_f32(noise, "noise")
        if tuple(noise.shape) != (geo.n_samples, geo.cin):
            raise ValueError("noise must be [N, Cin]")
    y = torch.empty(geo.n_samples, 3, dtype=torch.float32, device=g0.device)
    d = geo.to_desc(g0, g1, origins_aligned(geo, coord))
    pad = PaddedMlp.maybe(geo, params, False)
    if pad is not None:
        pad.pad()
        d.hidden = FUSED_HIDDEN
    m = pad.m if pad is not None else _mlp_struct(params)
    _lib.check(_lib.load().nic_fused_forward(ctypes.byref(d), _lib.ptr(g0), _lib.ptr(g1), _lib.ptr(org), ctypes.byref(m),
                                             _lib.ptr(noise if geo.noise_mode == NIC_NOISE_TENSOR else None), _lib.ptr(y),
                                             _lib.stream_ptr(g0.device)), "nic_fused_forward")
    return y


@_on_tensor_device
def fused_forward_u8(geo: PathGeometry, g0_u8, g1_u8, coord, params, out: str = "float"):
    """Decode from the STORED grids (the uint8 tensors ``fp_savable`` wrote, fp_def.py:250-255) without materialising fp32
    grids: dequantisation happens in the gather, bit-identical to ``fp_load`` + :func:`fused_forward`.
    ``out``: "float" -> fp32 [N,3]; "uint8" -> ``quantize_to_bit`` of it as bytes (models.py:39-40); "both" -> (fp32, uint8)."""
    for name, g in (("G0", g0_u8), ("G1", g1_u8)):
        if not isinstance(g, torch.Tensor) or g.dtype != torch.uint8:
            raise TypeError(f"{name} must be a uint8 tensor (fp_savable output)")
        if not g.is_cuda:
            raise RuntimeError(f"{name} lives on {g.device}: this package only runs on a HIP device (no CPU path)")
    g0_u8, g1_u8 = g0_u8.contiguous(), g1_u8.contiguous()
    check_grids(geo, g0_u8, g1_u8)
    if geo.noise_mode != NIC_NOISE_NONE:
        raise ValueError("decoding adds no noise")
    if out not in ("float", "uint8", "both"):
        raise ValueError("out must be 'float', 'uint8' or 'both'")
    params = check_mlp([p.detach() for p in params], geo.cin, geo.hidden)
    org = upload_origins(geo, coord, g0_u8.device, g0_u8, g1_u8)
    y = torch.empty(geo.n_samples, 3, dtype=torch.float32, device=g0_u8.device) if out != "uint8" else None
    yq = torch.empty(geo.n_samples, 3, dtype=torch.uint8, device=g0_u8.device) if out != "float" else None
    d = geo.to_desc(g0_u8, g1_u8, origins_aligned(geo, coord))
    pad = PaddedMlp.maybe(geo, params, False)
    if pad is not None:
        pad.pad()
        d.hidden = FUSED_HIDDEN
    m = pad.m if pad is not None else _mlp_struct(params)
    _lib.check(_lib.load().nic_fused_forward_u8(ctypes.byref(d), _lib.ptr(g0_u8), _lib.ptr(g1_u8), _lib.ptr(org), ctypes.byref(m),
                                                _lib.ptr(y), _lib.ptr(yq), _lib.stream_ptr(g0_u8.device)), "nic_fused_forward_u8")
    return y if out == "float" else (yq if out == "uint8" else (y, yq))


class TargetImage:
    """A resident image as the training target: pass it as ``target`` to :func:`fused_forward_backward` and
    sample i of crop k is compared with ``image[:, origin_k + i]`` - the reference's ``crop.reshape(3, -1).T`` stack
    (image_compression.py:37-47) is never built.  ``image``: the dataset tensor ``[3, S0, S1(, S2)]`` on the device, fp32, or
    uint8 codes with ``den`` (255: ToTensor, :436-440; 256: the 3D loader, :474; value = u / den, correctly rounded)."""

    def __init__(self, image: torch.Tensor, den: float = 255.0, rgbx: bool = False):
        """``rgbx``: ``image`` is the interleaved form ``[S0, S1(, S2)]`` int32 (R | G << 8 | B << 16 per sample, :func:`rgbx_interleave`):
        the kernel fetches a sample's three targets with one load."""
        self.rgbx = bool(rgbx)
        if self.rgbx:
            if not isinstance(image, torch.Tensor) or image.dtype != torch.int32 or image.dim() not in (2, 3):
                raise ValueError("an RGBX image is an int32 tensor [S0, S1] or [S0, S1, S2]")
        elif not isinstance(image, torch.Tensor) or image.dim() not in (3, 4) or image.shape[0] != 3:
            raise ValueError("image must be [3, S0, S1] or [3, S0, S1, S2]")
        if not image.is_cuda:
            raise RuntimeError(f"image lives on {image.device}: this package only runs on a HIP device (no CPU path)")
        if not self.rgbx and image.dtype not in (torch.float32, torch.uint8):
            raise NotImplementedError("resident images are fp32 or uint8")
        self.image = image.contiguous()
        self.den = float(den)

    @property
    def spatial(self):
        return tuple(self.image.shape) if self.rgbx else tuple(self.image.shape[1:])

    def to_struct(self, geo: "PathGeometry", coord) -> _lib.NicTargetImage:
        sp = self.spatial
        if len(sp) != geo.dim:
            raise ValueError("image / geometry dimension mismatch")
        if not (isinstance(coord, torch.Tensor) and coord.is_cuda):     # device origins are taken as is (no sync), like upload_origins
            o = torch.as_tensor(coord).reshape(-1, geo.dim).to(torch.int64)
            for a in range(geo.dim):
                if int(o[:, a].min()) < 0 or int(o[:, a].max()) + int(geo.extent[a]) > int(sp[a]):
                    raise IndexError(f"axis {a}: crop leaves the image ({int(o[:, a].max())} + {geo.extent[a]} > {sp[a]})")
        t = _lib.NicTargetImage()
        t.data = self.image.data_ptr()
        t.is_u8 = 2 if self.rgbx else (1 if self.image.dtype == torch.uint8 else 0)
        t.den = self.den
        for a in range(3):
            t.size[a] = int(sp[a]) if a < geo.dim else 1
        return t


@dataclass
class StepOutput:
    loss: torch.Tensor                       # 0-dim, the MSE mean (image_compression.py:259)
    y: Optional[torch.Tensor]
    grad_g0: torch.Tensor
    grad_g1: torch.Tensor
    grad_mlp: List[torch.Tensor]             # W1, b1, W2, b2, W3, b3
    flat: torch.Tensor                       # one buffer holding [loss | decoder grads | grid grads]


def grad_bucket_layout(geo: PathGeometry, g0: torch.Tensor, g1: torch.Tensor, n_linear: int = 3):
    """offsets (in floats) of [loss, W1, b1, ..., Wn, bn, G0, G1] inside one flat gradient buffer -
    one buffer so that data-parallel training needs a single all-reduce per step."""
    H, cin = geo.hidden, geo.cin
    sizes = [4, H * cin, H] + [H * H, H] * (n_linear - 2) + [3 * H, 3, g0.numel(), g1.numel()]      # loss padded to 16 B
    offs, o = [], 0
    for s in sizes:
        offs.append(o)
        o += (s + 3) // 4 * 4
    return offs, sizes, o


@_on_tensor_device
def fused_forward_backward(geo: PathGeometry, g0, g1, coord, params, target: torch.Tensor,
                           noise: Optional[torch.Tensor] = None, want_y: bool = False,
                           flat: Optional[torch.Tensor] = None, events=None, tail=None, clean: bool = False) -> StepOutput:
    """One training step's forward + MSE + backward in one launch (+ the fixed-order partial reduction):
    image_compression.py:239-265.  Gradients are returned, not accumulated into .grad.  ``tail``: callable ``(StepOutput views) -> optim.StepTail``
    or None: the optimiser step rides on the reduction launch (nic_path_desc.tail; ``optim.FusedAdam.step_tail``).  ``clean``: the caller vouches
    that the grid-gradient part of the reused ``flat`` is zero (the optimiser zeroed what it read, NIC_ADAM_ZERO_GRAD): no fill launch.  Grids may be bfloat16 / float16 STORAGE (2D,
    split_bf16): gathered values are widened to fp32, the returned grid gradients are fp32 tensors of the grids' shapes (feed them to
    ``optim.FusedAdam`` on fp32 masters with ``set_mirror``)."""
    g0 = _lib.require_cuda_grid(grid_storage(g0).detach(), "G0")
    g1 = _lib.require_cuda_grid(grid_storage(g1).detach(), "G1")
    check_grids(geo, g0, g1)
    params = check_mlp([p.detach() for p in params], geo.cin, geo.hidden)
    org = upload_origins(geo, coord, g0.device, g0, g1)
    timg = None
    if isinstance(target, TargetImage):
        timg = target.to_struct(geo, coord)
    else:
        target = _lib.require_cuda_f32(target, "target").reshape(-1, 3)
        if target.shape[0] != geo.n_samples:
            raise ValueError(f"target has {target.shape[0]} rows, geometry has {geo.n_samples} samples")
    if geo.noise_mode == NIC_NOISE_TENSOR:
        noise = _lib.require_cuda_f32(noise, "noise")
        if tuple(noise.shape) != (geo.n_samples, geo.cin):
            raise ValueError("noise must be [N, Cin]")
    dev = g0.device
    nl = len(params) // 2
    offs, sizes, total = grad_bucket_layout(geo, g0, g1, nl)
    if flat is None:
        flat = torch.zeros(total, dtype=torch.float32, device=dev)       # grid grads must start at zero
    else:
        if flat.numel() != total or flat.dtype != torch.float32 or not flat.is_cuda:
            raise ValueError("flat gradient buffer has the wrong size / dtype / device")
        if not clean:
            flat.zero_()
    views = [flat[o:o + s] for o, s in zip(offs, sizes)]
    shapes = [p.shape for p in params]
    gm = [views[1 + i].view(shapes[i]) for i in range(2 * nl)]
    gg0, gg1 = views[1 + 2 * nl].view(g0.shape), views[2 + 2 * nl].view(g1.shape)
    y = torch.empty(geo.n_samples, 3, dtype=torch.float32, device=dev) if want_y else None
    d = geo.to_desc(g0, g1, origins_aligned(geo, coord))
    lib = _lib.load()
    ws_bytes = int(lib.nic_workspace_bytes(ctypes.byref(d)))
    ws = _lib.workspace(dev, ws_bytes)
    pad = PaddedMlp.maybe(geo, params, True)
    if pad is not None:
        pad.pad()
        d.hidden = FUSED_HIDDEN
    m = pad.m if pad is not None else _mlp_struct(params)
    gs = pad.gs if pad is not None else _grads_struct(gm)
    th = None
    if tail is not None and pad is None:     # (zero-padded decoders reduce into padded copies: their step stays a launch of its own)
        th = tail(gg0, gg1, gm)
        if th is not None:
            d.tail = th.struct_ptr
    if events is not None:                   # (start, end) torch.cuda.Event pair recorded on the launch stream
        events[0].record(torch.cuda.current_stream(dev))
    noise_p = _lib.ptr(noise if geo.noise_mode == NIC_NOISE_TENSOR else None)
    if timg is not None:
        _lib.check(lib.nic_fused_forward_backward_img(
            ctypes.byref(d), _lib.ptr(g0), _lib.ptr(g1), _lib.ptr(org), ctypes.byref(m), noise_p, ctypes.byref(timg), _lib.ptr(y),
            _lib.ptr(views[0]), _lib.ptr(gg0), _lib.ptr(gg1), ctypes.byref(gs), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)),
            "nic_fused_forward_backward_img")
    else:
        _lib.check(lib.nic_fused_forward_backward(
            ctypes.byref(d), _lib.ptr(g0), _lib.ptr(g1), _lib.ptr(org), ctypes.byref(m), noise_p, _lib.ptr(target), _lib.ptr(y),
            _lib.ptr(views[0]), _lib.ptr(gg0), _lib.ptr(gg1), ctypes.byref(gs), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)),
            "nic_fused_forward_backward")
    if th is not None:
        th.commit()
    if pad is not None:
        pad.unpad(gm)
    if events is not None:
        events[1].record(torch.cuda.current_stream(dev))
    return StepOutput(views[0][0], y, gg0, gg1, gm, flat)


class StepPlan:
    """``fused_forward_backward`` for a training loop: everything that does not change from step to step - the checked grids and
    decoder parameters, the descriptor, the gradient bucket and its views, the parameter / gradient structs - is prepared once; ``run`` rewrites the noise fields and the origins and launches.  The Python side
    of a step drops from ~60 to ~15 us (the reference's default step is host-bound: 8 x 256^2 samples take the GPU 0.21 ms).
    Targets: a resident :class:`TargetImage`, or a resident fp32 ``[N, 3]`` tensor (the reference's crop stack, image_compression.py:37-47; the same rows
    every step - whole-domain passes).  Origins: a host tensor / list (validated) or a device int32 tensor (taken as is)."""

    LOSS_SLOTS = 1 << 16

    def __init__(self, geo: PathGeometry, g0, g1, params, target):
        if not isinstance(target, TargetImage):
            target = _lib.require_cuda_f32(target, "target").reshape(-1, 3)
            if target.shape[0] != geo.n_samples:
                raise ValueError(f"target has {target.shape[0]} rows, geometry has {geo.n_samples} samples")
        self.geo = geo
        self.g0 = _lib.require_cuda_grid(grid_storage(g0).detach(), "G0")
        self.g1 = _lib.require_cuda_grid(grid_storage(g1).detach(), "G1")
        check_grids(geo, self.g0, self.g1)
        self.params = check_mlp([p.detach() for p in params], geo.cin, geo.hidden)
        self.dev = self.g0.device
        nl = len(self.params) // 2
        offs, sizes, total = grad_bucket_layout(geo, self.g0, self.g1, nl)
        self.flat = torch.zeros(total, dtype=torch.float32, device=self.dev)
        views = [self.flat[o:o + s] for o, s in zip(offs, sizes)]
        # the loss of step k goes to slot k % LOSS_SLOTS of a buffer of its own: the value a step returns stays valid without a copy out of
        # the reused bucket (the copy kernel was 4.7 us of a 218 us step).  When the slots run out the plan moves on to a FRESH buffer: the
        # views handed out so far keep the old one alive, so a loss history of any length stays correct (the reference's launchers run
        # NUM_EPOCHS = 320 000 on one plan; 256 KB per 65 536 steps)
        self.loss_buf = torch.zeros(self.LOSS_SLOTS, dtype=torch.float32, device=self.dev)
        self.loss_base = self.loss_buf.data_ptr()
        self.steps = 0
        self.clean = False                   # True: the grid-gradient part of the bucket is known to be zero (the decoder part is overwritten)
        self.gm = [views[1 + i].view(self.params[i].shape) for i in range(2 * nl)]
        self.gg0, self.gg1 = views[1 + 2 * nl].view(self.g0.shape), views[2 + 2 * nl].view(self.g1.shape)
        self.d = geo.to_desc(self.g0, self.g1, False)
        self.pad = PaddedMlp.maybe(geo, self.params, True)           # hidden widths below 64: the kernel reads zero-padded copies
        if self.pad is not None:
            self.d.hidden = FUSED_HIDDEN
        self.m = self.pad.m if self.pad is not None else _mlp_struct(self.params)
        self.gs = self.pad.gs if self.pad is not None else _grads_struct(self.gm)
        self.timg = target.to_struct(geo, [[0] * geo.dim] * geo.num_crops) if isinstance(target, TargetImage) else None
        self.target = target
        self.lib = _lib.load()
        self.n_org = geo.num_crops * geo.dim
        # host-side bounds of a valid origin (check_origins + TargetImage.to_struct, per axis)
        self.hi = []
        sp = target.spatial if self.timg is not None else None
        for a in range(geo.dim):
            s = float(geo.step_number)
            n0, n1 = int(self.g0.shape[-(a + 1)]), int(self.g1.shape[-(a + 1)])
            ext = int(geo.extent[a])
            if sp is not None:
                hi = int(sp[a]) - ext
            else:                                     # no image: the grids bound the origin (last sample t: floor(t s) + 1 <= n0 - 1, floor(t s / 2) + 1 <= n1 - 1)
                hi = min(math.ceil((n0 - 1) / s), math.ceil((n1 - 1) * 2 / s)) - ext
            while hi >= 0 and (math.floor((hi + ext - 1) * s) + 1 > n0 - 1 or math.floor((hi + ext - 1) * s / 2) + 1 > n1 - 1):
                hi -= 1
            self.hi.append(hi)

    def matches(self, g0, g1, params, target) -> bool:
        return (grid_storage(g0).data_ptr() == self.g0.data_ptr() and grid_storage(g1).data_ptr() == self.g1.data_ptr() and target is self.target
                and len(params) == len(self.params) and all(p.data_ptr() == q.data_ptr() for p, q in zip(params, self.params)))

    def run(self, coord, noise_mode: int, noise_seed: int, noise_offset: int, tail=None, events=None) -> StepOutput:
        """``tail``: an ``optim.StepTail`` built on this plan's buffers (``FusedAdam.step_tail([(g0, plan.gg0), (g1, plan.gg1)], zip(params, plan.gm))``):
        the optimiser step rides on the reduction launch; committed here once the launch is queued"""
        geo, dev = self.geo, self.dev
        with torch.cuda.device(dev):
            if isinstance(coord, torch.Tensor) and coord.is_cuda:
                org = coord.reshape(-1).to(torch.int32)
                if org.numel() != self.n_org:
                    raise ValueError(f"need {geo.num_crops} origins")
            else:
                host = torch.as_tensor(coord).reshape(-1, geo.dim)
                if host.shape[0] != geo.num_crops:
                    raise ValueError(f"need {geo.num_crops} origins, got {host.shape[0]}")
                rows = host.tolist()          # plain Python from here: a dim-wise min / max of this 8 x 2 CPU tensor wakes torch's intra-op
                for a in range(geo.dim):      # thread pool and stalls ~90 ms every ~50 calls on the 128-thread GPU boxes
                    col = [r[a] for r in rows]
                    if min(col) < 0 or max(col) > self.hi[a]:
                        raise IndexError(f"axis {a}: crop origin {min(col)}..{max(col)} outside [0, {self.hi[a]}] (image / grid bounds)")
                if geo.num_crops <= _lib.NIC_ORIGINS_INLINE_MAX and os.environ.get("NIC_NO_HOST_ORIGINS") != "1":
                    # host values ride in the kernel arguments (NIC_FLAG_ORIGINS_HOST): no upload, no dependent global load in front of the gathers
                    org = (ctypes.c_int32 * self.n_org)(*[int(v) for r in rows for v in r])
                else:
                    org = host.to(torch.int32).reshape(-1).to(dev, non_blocking=True)     # (a pinned staging ring + async copies measured 10 - 100 x slower here)
            slot = self.steps % self.LOSS_SLOTS
            if slot == 0 and self.steps > 0:
                self.loss_buf = torch.zeros(self.LOSS_SLOTS, dtype=torch.float32, device=self.dev)
                self.loss_base = self.loss_buf.data_ptr()
            self.steps += 1
            d = self.d
            d.noise_mode = int(noise_mode)
            d.noise_seed = int(noise_seed) & 0xFFFFFFFFFFFFFFFF
            d.noise_offset = int(noise_offset) & 0xFFFFFFFFFFFFFFFF
            if not self.clean:
                self.flat.zero_()
            self.clean = False               # set again by whoever zeroes the grid gradients (FusedAdam.zero_grad_in_step: in its own launch)
            ws = _lib.workspace(dev, int(self.lib.nic_workspace_bytes(ctypes.byref(d))))
            if self.pad is not None:
                self.pad.pad()
                tail = None                  # zero-padded decoders reduce into padded copies: their step stays a launch of its own
            d.tail = tail.struct_ptr if tail is not None else None
            host_org = not isinstance(org, torch.Tensor)
            org_p = ctypes.cast(org, ctypes.c_void_p) if host_org else _lib.ptr(org)
            flags0 = d.flags
            if host_org:
                d.flags = flags0 | _lib.NIC_FLAG_ORIGINS_HOST
            if events is not None:           # (start, end) event pair recorded on the launch stream (bench.KernelEvents brackets the fused kernel alone)
                events[0].record(torch.cuda.current_stream(dev))
            try:
                if self.timg is not None:
                    _lib.check(self.lib.nic_fused_forward_backward_img(
                        ctypes.byref(d), _lib.ptr(self.g0), _lib.ptr(self.g1), org_p, ctypes.byref(self.m), None, ctypes.byref(self.timg), None,
                        self.loss_base + 4 * slot, _lib.ptr(self.gg0), _lib.ptr(self.gg1), ctypes.byref(self.gs), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)),
                        "nic_fused_forward_backward_img")
                else:
                    _lib.check(self.lib.nic_fused_forward_backward(
                        ctypes.byref(d), _lib.ptr(self.g0), _lib.ptr(self.g1), org_p, ctypes.byref(self.m), None, _lib.ptr(self.target), None,
                        self.loss_base + 4 * slot, _lib.ptr(self.gg0), _lib.ptr(self.gg1), ctypes.byref(self.gs), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)),
                        "nic_fused_forward_backward")
            finally:
                d.tail = None
                d.flags = flags0
            if events is not None:
                events[1].record(torch.cuda.current_stream(dev))
            if tail is not None:
                tail.commit()
            if self.pad is not None:
                self.pad.unpad(self.gm)
        return StepOutput(self.loss_buf[slot], None, self.gg0, self.gg1, self.gm, self.flat)

    def launch_dev(self, origins_dev: torch.Tensor, step_dev_ptr: int, loss_slot: torch.Tensor, ws: torch.Tensor, noise_mode: int, noise_seed: int,
                   noise_base: int = 0, tail_struct=None) -> None:
        """the step of a hipGraph-captured loop (``nic_fused_forward_backward_img_dev``): origins from a device buffer, the step number from
        device memory (added to ``noise_base``), the loss into ``loss_slot``.  Touches no host state and allocates nothing: capture-safe; the
        grid-gradient part of the bucket must be clean (``optim.FusedAdam`` zeroes it in its own launch, NIC_ADAM_ZERO_GRAD)."""
        if self.timg is None:
            raise TypeError("the captured step reads its targets from a resident TargetImage")
        d = self.d
        d.noise_mode = int(noise_mode)
        d.noise_seed = int(noise_seed) & 0xFFFFFFFFFFFFFFFF
        d.noise_offset = int(noise_base) & 0xFFFFFFFFFFFFFFFF
        if self.pad is not None:
            self.pad.pad()
        # tail_struct: a NicStepTail with the device schedule (optim.DevAdam.tail_struct) - the optimiser rides on the reduction launch of the captured step
        d.tail = ctypes.addressof(tail_struct) if (tail_struct is not None and self.pad is None) else None
        try:
            _lib.check(self.lib.nic_fused_forward_backward_img_dev(
                ctypes.byref(d), _lib.ptr(self.g0), _lib.ptr(self.g1), _lib.ptr(origins_dev), ctypes.byref(self.m), ctypes.byref(self.timg),
                _lib.ptr(loss_slot), _lib.ptr(self.gg0), _lib.ptr(self.gg1), ctypes.byref(self.gs), ctypes.c_void_p(step_dev_ptr), _lib.ptr(ws), ws.numel(),
                _lib.stream_ptr(self.dev)), "nic_fused_forward_backward_img_dev")
        finally:
            d.tail = None
        if self.pad is not None:
            self.pad.unpad(self.gm)


# ------------------------------------------------------------------------------------------------------
# autograd
# ------------------------------------------------------------------------------------------------------

# ---------------------------------------------------------------------------------------------------------------------------------------
# multi-level fused step (nic_fused_ml_*: several level pairs per sample, csrc/fused_q16.hpp::QML)
# ---------------------------------------------------------------------------------------------------------------------------------------
ML_FUSED = {(2, 4, 3), (3, 4, 3), (5, 4, 3), (2, 4, 5), (3, 4, 5), (2, 12, 3), (3, 12, 3)}      # (levels, C, n_linear) with P = 6, H = 64: what fits the LDS


def ml_is_fused(levels: int, channels: int, pe_channels: int, hidden: int, n_linear: int) -> bool:
    return (int(levels), int(channels), int(n_linear)) in ML_FUSED and int(pe_channels) == 6 and int(hidden) == 64


def _ml_pairs(fp: Sequence[torch.Tensor], grads: Optional[Sequence[torch.Tensor]] = None) -> "_lib.NicMlPairs":
    if len(fp) % 2 or not 2 <= len(fp) // 2 <= _lib.NIC_ML_MAX_LEVELS:
        raise ValueError(f"2 .. {_lib.NIC_ML_MAX_LEVELS} level pairs")
    pr = _lib.NicMlPairs()
    pr.levels = len(fp) // 2
    for l in range(pr.levels):
        a, b = fp[2 * l], fp[2 * l + 1]
        pr.g0[l], pr.g1[l] = a.data_ptr(), b.data_ptr()
        for ax in range(2):
            pr.g0_nodes[l][ax] = int(a.shape[-(ax + 1)])
            pr.g1_nodes[l][ax] = int(b.shape[-(ax + 1)])
        if grads is not None:
            pr.g0_grad[l], pr.g1_grad[l] = grads[2 * l].data_ptr(), grads[2 * l + 1].data_ptr()
    return pr


@dataclasses.dataclass
class MlStepOutput:
    loss: torch.Tensor                    # device scalar
    y: Optional[torch.Tensor]
    grad_fp: List[torch.Tensor]           # one dense fp32 gradient per grid, in fp order
    grad_mlp: List[torch.Tensor]


def _ml_geo_desc(geo: PathGeometry, fp, coord) -> "_lib.NicPathDesc":
    if geo.dim != 2 or geo.method != 1:
        raise NotImplementedError("the multi-level mode is 2D")
    return geo.to_desc(fp[0], fp[1], origins_aligned(geo, coord))


@_on_tensor_device
def fused_ml_forward_backward(geo: PathGeometry, fp: Sequence[torch.Tensor], coord, params, target: torch.Tensor, noise: Optional[torch.Tensor] = None,
                              want_y: bool = False, grads: Optional[Sequence[torch.Tensor]] = None, mlp_grads: Optional[Sequence[torch.Tensor]] = None,
                              loss: Optional[torch.Tensor] = None, events=None, tail=None) -> MlStepOutput:
    """One multi-level training step in ONE launch (+ the fixed-order reduction of the decoder-gradient records): ``geo`` describes pair 0
    (``step_number`` = 2^(mip - 2): pair l runs at 4^-l of it), ``fp`` = [G0_0, G1_0, G0_1, G1_1, ..] fp32 grids.  ``grads`` (one fp32 tensor per grid) are
    ADDED to when given - a whole-image pass walked in chunks - and freshly zeroed otherwise; decoder gradients and the loss are overwritten.
    Raises ``_lib.Unsupported`` for (levels, C, depth) combinations without a fused kernel (``ml_is_fused``)."""
    fp = [_lib.require_cuda_f32(g.detach(), f"fp[{i}]") for i, g in enumerate(fp)]
    L = len(fp) // 2
    cin = L * (5 * geo.channels + 2 * geo.pe_channels) + 1
    params = check_mlp([p.detach() for p in params], cin, geo.hidden)
    dev = fp[0].device
    org = upload_origins(geo, coord, dev, fp[0], fp[1])
    target = _lib.require_cuda_f32(target, "target").reshape(-1, 3)
    if target.shape[0] != geo.n_samples:
        raise ValueError(f"target has {target.shape[0]} rows, geometry has {geo.n_samples} samples")
    if geo.noise_mode == NIC_NOISE_TENSOR:
        noise = _lib.require_cuda_f32(noise, "noise")
        if tuple(noise.shape) != (geo.n_samples, cin):
            raise ValueError("noise must be [N, Cin]")
    if grads is None:
        grads = [torch.zeros_like(g) for g in fp]
    if mlp_grads is None:
        mlp_grads = [torch.empty_like(p) for p in params]
    if loss is None:
        loss = torch.zeros(1, dtype=torch.float32, device=dev)
    y = torch.empty(geo.n_samples, 3, dtype=torch.float32, device=dev) if want_y else None
    d = _ml_geo_desc(geo, fp, coord)
    d.flags &= ~(_lib.NIC_FLAG_SPLIT_BF16 | _lib.NIC_FLAG_BF16)
    pr = _ml_pairs(fp, grads)
    lib = _lib.load()
    ws = _lib.workspace(dev, int(lib.nic_workspace_bytes(ctypes.byref(d))))
    m, gs = _mlp_struct(params), _grads_struct(mlp_grads)
    if tail is not None:                       # optim.StepTail on `grads` / `mlp_grads`: the optimiser step rides on the reduction launch
        d.tail = tail.struct_ptr
    if events is not None:
        events[0].record(torch.cuda.current_stream(dev))
    _lib.check(lib.nic_fused_ml_forward_backward(ctypes.byref(d), ctypes.byref(pr), _lib.ptr(org), ctypes.byref(m),
                                                 _lib.ptr(noise if geo.noise_mode == NIC_NOISE_TENSOR else None), _lib.ptr(target), _lib.ptr(y), _lib.ptr(loss),
                                                 ctypes.byref(gs), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)), "nic_fused_ml_forward_backward")
    if tail is not None:
        tail.commit()
    if events is not None:
        events[1].record(torch.cuda.current_stream(dev))
    return MlStepOutput(loss[0], y, list(grads), list(mlp_grads))


@_on_tensor_device
def fused_ml_forward(geo: PathGeometry, fp: Sequence[torch.Tensor], coord, params) -> torch.Tensor:
    """forward pass of the multi-level field for the crops at ``coord``: [N, 3]"""
    fp = [_lib.require_cuda_f32(g.detach(), f"fp[{i}]") for i, g in enumerate(fp)]
    L = len(fp) // 2
    cin = L * (5 * geo.channels + 2 * geo.pe_channels) + 1
    params = check_mlp([p.detach() for p in params], cin, geo.hidden)
    dev = fp[0].device
    org = upload_origins(geo, coord, dev, fp[0], fp[1])
    y = torch.empty(geo.n_samples, 3, dtype=torch.float32, device=dev)
    d = _ml_geo_desc(geo, fp, coord)
    d.flags &= ~(_lib.NIC_FLAG_SPLIT_BF16 | _lib.NIC_FLAG_BF16)
    d.noise_mode = NIC_NOISE_NONE
    pr = _ml_pairs(fp)
    m = _mlp_struct(params)
    _lib.check(_lib.load().nic_fused_ml_forward(ctypes.byref(d), ctypes.byref(pr), _lib.ptr(org), ctypes.byref(m), _lib.ptr(y), _lib.stream_ptr(dev)),
               "nic_fused_ml_forward")
    return y


class FusedGridMLP(torch.autograd.Function):
    """y[N,3] = decoder(encode(G0, G1) + noise) as ONE differentiable op.  Inputs are the caller's own leaf
    tensors (the reference keeps the grids as raw leaves in a Python list registered in Adam,
    image_compression.py:361-364), read by pointer on every call.  backward() recomputes the forward inside
    the backward kernel and returns dense gradients of the grids' shapes."""

    @staticmethod
    def forward(ctx, g0, g1, geo: PathGeometry, org: torch.Tensor, noise, *params):
        params = list(params)
        y = fused_forward(geo, g0, g1, org, params, noise)
        ctx.save_for_backward(g0, g1, *params)
        ctx.geo, ctx.org, ctx.noise = geo, org, noise
        return y

    @staticmethod
    @_on_tensor_device
    def backward(ctx, dy):
        g0, g1, *params = ctx.saved_tensors
        geo = ctx.geo
        g0c = _lib.require_cuda_f32(g0.detach(), "G0")
        g1c = _lib.require_cuda_f32(g1.detach(), "G1")
        pc = check_mlp([p.detach() for p in params], geo.cin, geo.hidden)
        dy = _lib.require_cuda_f32(dy, "dy")
        dev = g0c.device
        gg0, gg1 = torch.zeros_like(g0c), torch.zeros_like(g1c)
        gm = [torch.empty_like(p) for p in pc]
        d = geo.to_desc(g0c, g1c)
        if geo.fp16 and geo.dz_scale_log2 == 0:
            # fp16 products carry 2^k dZ: an upstream mean-squared error hands in dY = 2 (y - t) / (3 N) - k brings that to O(1)
            # (pass PathGeometry.dz_scale_log2 for any other loss)
            d.dz_scale_log2 = int(round(math.log2(3.0 * geo.n_samples))) + 1
        lib = _lib.load()
        ws = _lib.workspace(dev, int(lib.nic_workspace_bytes(ctypes.byref(d))))
        pad = PaddedMlp.maybe(geo, pc, True)
        if pad is not None:
            pad.pad()
            d.hidden = FUSED_HIDDEN
        m, gs = (pad.m, pad.gs) if pad is not None else (_mlp_struct(pc), _grads_struct(gm))
        noise = ctx.noise if geo.noise_mode == NIC_NOISE_TENSOR else None
        _lib.check(lib.nic_fused_backward_dy(ctypes.byref(d), _lib.ptr(g0c), _lib.ptr(g1c), _lib.ptr(ctx.org), ctypes.byref(m),
                                             _lib.ptr(noise), _lib.ptr(dy), _lib.ptr(gg0), _lib.ptr(gg1), ctypes.byref(gs),
                                             _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)), "nic_fused_backward_dy")
        if pad is not None:
            pad.unpad(gm)
        need = ctx.needs_input_grad
        return (gg0 if need[0] else None, gg1 if need[1] else None, None, None, None, *[gm[i] if need[5 + i] else None for i in range(len(gm))])


def fused_grid_mlp(geo: PathGeometry, g0, g1, coord, params, noise: Optional[torch.Tensor] = None) -> torch.Tensor:
    """differentiable fused op (see FusedGridMLP)"""
    check_grids(geo, g0, g1)
    org = upload_origins(geo, coord, g0.device, g0, g1)
    if geo.noise_mode == NIC_NOISE_TENSOR:
        noise = _lib.require_cuda_f32(noise, "noise")
    if origins_aligned(geo, coord):
        geo = dataclasses.replace(geo, flags=geo.flags | _lib.NIC_FLAG_ORIGINS_ALIGNED)
    return FusedGridMLP.apply(g0, g1, geo, org, noise, *params)


class EncodeFunction(torch.autograd.Function):
    """create_decoder_input_* as a differentiable op: forward nic_encode, backward nic_encode_backward (dense grid
    gradients, like autograd through the reference's advanced-indexing gathers)."""

    @staticmethod
    def forward(ctx, g0, g1, geo: PathGeometry, org: torch.Tensor):
        ctx.geo, ctx.org = geo, org
        ctx.shapes = (g0.shape, g1.shape)
        ctx.dev = g0.device
        return encode(geo, g0, g1, org)

    @staticmethod
    @_on_tensor_device
    def backward(ctx, dx):
        geo = ctx.geo
        dx = _lib.require_cuda_f32(dx, "dx")
        gg0 = torch.zeros(ctx.shapes[0], dtype=torch.float32, device=ctx.dev)
        gg1 = torch.zeros(ctx.shapes[1], dtype=torch.float32, device=ctx.dev)
        d = geo.to_desc(gg0, gg1)
        _lib.check(_lib.load().nic_encode_backward(ctypes.byref(d), _lib.ptr(ctx.org), _lib.ptr(dx), _lib.ptr(gg0), _lib.ptr(gg1),
                                                   _lib.stream_ptr(ctx.dev)), "nic_encode_backward")
        return gg0, gg1, None, None


def encode_differentiable(geo: PathGeometry, g0, g1, coord) -> torch.Tensor:
    check_grids(geo, g0, g1)
    org = upload_origins(geo, coord, g0.device, g0, g1)
    if not (g0.requires_grad or g1.requires_grad):
        return encode(geo, g0, g1, org)
    return EncodeFunction.apply(g0, g1, geo, org)


def decoder_is_specialised(cin: int, hidden: int, n_linear: int) -> bool:
    """the reference's defaults (Cin of the three layouts, H = 64, 3 Linear layers) run on the MFMA decoder kernel; every other width /
    depth on the layer-wise general kernels (csrc/decoder_general.hip)"""
    return cin in (73, 127, 79) and hidden == 64 and n_linear == 3


class DecoderFunction(torch.autograd.Function):
    """ColorDecoder.forward on an explicit [n, Cin] input (image_compression.py:66-68) and its backward; ``params`` = W1, b1, W2, b2, ..
    in nn.Sequential order, any Cin / HIDDEN_LAYER_CHANNELS, 2 .. 5 Linear layers."""

    @staticmethod
    @_on_tensor_device
    def forward(ctx, x, *params):
        hidden = params[0].shape[0]
        params = check_mlp(params, x.shape[1], hidden)
        xc = _lib.require_cuda_f32(x, "x")
        n, cin = xc.shape
        nl = len(params) // 2
        y = torch.empty(n, 3, dtype=torch.float32, device=xc.device)
        m = _mlp_struct(params)
        lib = _lib.load()
        if decoder_is_specialised(cin, hidden, nl):
            _lib.check(lib.nic_decoder_forward(ctypes.byref(m), _lib.ptr(xc), n, cin, hidden, _lib.ptr(y), _lib.stream_ptr(xc.device)), "nic_decoder_forward")
        else:
            ws = _lib.workspace(xc.device, int(lib.nic_decoder_general_workspace_bytes(n, cin, hidden, nl, 0)))
            _lib.check(lib.nic_decoder_general_forward(ctypes.byref(m), _lib.ptr(xc), n, cin, hidden, _lib.ptr(y), _lib.ptr(ws), ws.numel(),
                                                       _lib.stream_ptr(xc.device)), "nic_decoder_general_forward")
        ctx.save_for_backward(xc, *params)
        return y

    @staticmethod
    @_on_tensor_device
    def backward(ctx, dy):
        x, *params = ctx.saved_tensors
        dy = _lib.require_cuda_f32(dy, "dy")
        n, cin = x.shape
        hidden, nl = params[0].shape[0], len(params) // 2
        dev = x.device
        need = ctx.needs_input_grad
        dx = torch.empty_like(x) if need[0] else None
        gm = [torch.empty_like(p) for p in params]
        lib = _lib.load()
        m, gs = _mlp_struct(params), _grads_struct(gm)
        if n == 0:
            for g in gm:
                g.zero_()
        elif decoder_is_specialised(cin, hidden, nl):
            ws = _lib.workspace(dev, int(lib.nic_workspace_bytes(None)))
            _lib.check(lib.nic_decoder_backward(ctypes.byref(m), _lib.ptr(x), _lib.ptr(dy), n, cin, hidden, _lib.ptr(dx),
                                                ctypes.byref(gs), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)), "nic_decoder_backward")
        else:
            ws = _lib.workspace(dev, int(lib.nic_decoder_general_workspace_bytes(n, cin, hidden, nl, 1)))
            _lib.check(lib.nic_decoder_general_backward(ctypes.byref(m), _lib.ptr(x), _lib.ptr(dy), n, cin, hidden, _lib.ptr(dx),
                                                        ctypes.byref(gs), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)), "nic_decoder_general_backward")
        return (dx, *[gm[i] if need[1 + i] else None for i in range(len(gm))])
