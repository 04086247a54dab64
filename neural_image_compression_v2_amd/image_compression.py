"""Driver-level call surface of the reference's Projects/image_compression.py for the hot path:
ColorDecoder, create_decoder_input_2d/_3d/_3d_v2, finally_decode_input_*, random_crop_dataset, train_models,
decode_image - with the same names, argument meaning and channel / sample orders, running on the HIP library.

The reference is a script that reads module globals from var2.py; here the same names live in a
``var2.Settings`` object held by ``ImageCompression``, whose methods keep the reference signatures.
``train_models`` offers the reference's unfused sequence (differentiable encode -> noise -> decoder -> MSELoss ->
backward; every op a HIP kernel of this package) and the fused single-launch step (default).
"""
from __future__ import annotations

import math
import ctypes
import os
import random
import time
from typing import List, Optional, Sequence

import torch
import torch.nn as nn

from . import _lib, fused
from .fp_def import (create_pyramid, create_pyramid_3d, create_pyramid_mip_levels, fp_all_quantize, fp_freeze,
                     fp_quantize_clamp)
from .models import quantize_to_bit
from .optim import CosineAnnealing, FusedAdam
from .utils import calculate_psnr
from .var2 import Settings


class ColorDecoder(nn.Module):
    """Linear(Cin,H)-GELU-Linear(H,H)-GELU-Linear(H,3)-Sigmoid (image_compression.py:54-68).  The nn.Sequential only
    owns the parameters (state_dict keys decoder.{0,2,4}.{weight,bias}, default nn.Linear init); forward() runs the
    whole stack in HIP kernels with their own autograd backward: one MFMA kernel (nic_decoder_forward / _backward) for the reference's
    defaults (Cin 73 / 127 / 79, H = 64, 3 layers), the layer-wise general kernels (nic_decoder_general_*) for every other
    DECODER_INPUT_CHANNELS / HIDDEN_LAYER_CHANNELS (var2.py:72,114-118) and depth."""

    def __init__(self, decoder_input_channels: int = 73, hidden_layer_channels: int = 64, n_linear: int = 3):
        """``n_linear``: 3 = the reference's decoder (depth is hard-coded there); 5 = the "4 x 64" decoder of the north star
        (keys decoder.{0,2,4,6,8}); 2 .. 5 accepted."""
        super().__init__()
        if not 2 <= n_linear <= _lib.NIC_MAX_LINEAR:
            raise NotImplementedError(f"2 .. {_lib.NIC_MAX_LINEAR} Linear layers")
        layers = [nn.Linear(decoder_input_channels, hidden_layer_channels), nn.GELU()]
        for _ in range(n_linear - 2):
            layers += [nn.Linear(hidden_layer_channels, hidden_layer_channels), nn.GELU()]
        layers += [nn.Linear(hidden_layer_channels, 3), nn.Sigmoid()]
        self.decoder = nn.Sequential(*layers)
        self.n_linear = n_linear

    def linear_params(self) -> List[torch.Tensor]:
        return [t for m in self.decoder if isinstance(m, nn.Linear) for t in (m.weight, m.bias)]

    def forward(self, x):
        return fused.DecoderFunction.apply(x, *self.linear_params())


class ImageCompression:
    """State + methods of the reference script for one fit."""

    def __init__(self, cfg: Optional[Settings] = None, device=None, seed: Optional[int] = None):
        self.cfg = cfg or Settings()
        c = self.cfg
        self.device = torch.device(device) if device is not None else c.DEVICE
        if self.device.type != "cuda":
            raise RuntimeError("ImageCompression needs a HIP device: there is no CPU implementation of this path")
        if c.MLP_NUM_DTYPE not in (16, 32):
            raise NotImplementedError("MLP_NUM_DTYPE is 32, or 16 = float16 grid storage (utils.py:301-313 maps 16 to torch.float16; the reference "
                                      "never casts its decoder, image_compression.py:350, and its own 16-bit run does not train, readme.md:9)")
        # (FEATURE_PYRAMID_CHANNELS / PE_CHANNELS other than 12 / 6: the element-wise API path takes any C; the FUSED step and decode exist for
        #  C in 4, 8, 12, 16 / P in 4, 6, 8 on the plain-bf16 kernels - TF_PLAIN_BF16=True - and raise NIC_E_UNSUPPORTED otherwise)
        grid_dtype = torch.bfloat16 if c.TF_GRID_BF16 else (torch.float16 if c.MLP_NUM_DTYPE == 16 else torch.float32)
        if grid_dtype != torch.float32 and c.FP_DIMENSION == 3 and not c.plain_bf16:
            raise NotImplementedError("16-bit grid storage in 3D runs on the plain-bf16 kernels: set TF_PLAIN_BF16=True")
        if grid_dtype != torch.float32 and c.FP_DIMENSION == 2 and not (c.TF_SPLIT_BF16 or c.plain_bf16):
            raise NotImplementedError("16-bit grid storage needs TF_SPLIT_BF16 or TF_PLAIN_BF16")
        if seed is not None:
            torch.manual_seed(seed)
            random.seed(seed)
        self.decoder = ColorDecoder(c.DECODER_INPUT_CHANNELS, c.HIDDEN_LAYER_CHANNELS, c.DECODER_LINEAR_LAYERS).to(self.device)     # :350
        pyr = create_pyramid if c.FP_DIMENSION == 2 else create_pyramid_3d
        self.feature_pyramid, self.feature_pyramid_levels = pyr(c.FEATURE_PYRAMID_SIZE, c.FEATURE_PYRAMID_CHANNELS, c.FP_BITS,
                                                                 self.device, grid_dtype, c.TF_NO_MIP)       # :352-357
        self.feature_pyramid_mip_levels_dict = create_pyramid_mip_levels(c.IMAGE_SIZE, c.FEATURE_PYRAMID_SIZE)  # :360
        # Adam + the grids' clamp in one launch per step (optim.FusedAdam; state layout of torch.optim.Adam)
        self.optimizer = FusedAdam([{"params": self.feature_pyramid, "lr": 0.01},
                                    {"params": self.decoder.parameters(), "lr": 0.005}])                   # :361-364
        self.optimizer.set_clamp(self.feature_pyramid, -(2 ** c.FP_BITS - 1) / 2 ** (c.FP_BITS + 1), 0.5)  # fp_def.py:227-232
        self.optimizer.zero_grad_in_step(self.feature_pyramid)      # the grids' gradient buckets are zeroed by the Adam launch itself
        for t in self.feature_pyramid:                              # 16-bit storage: the launch that updates a master rewrites its mirror
            if getattr(t, "mirror16", None) is not None:
                self.optimizer.set_mirror(t, t.mirror16)
        self.scheduler = CosineAnnealing(self.optimizer, T_max=c.NUM_EPOCHS, eta_min=0)   # :365 (torch's CosineAnnealingLR, bit for bit, without its overhead)
        self.images: List[torch.Tensor] = []
        self.loss_history: List[torch.Tensor] = []
        self.step_count = 0
        if c.TF_PRINT_LOG and os.environ.get("NIC_QUIET") != "1":
            import sys
            print(f"[nicv2] product arithmetic of the fused steps: {self.arithmetic}", file=sys.stderr, flush=True)

    @property
    def arithmetic(self) -> str:
        """the resolved arithmetic of the fused training steps and decodes of this configuration (Settings.TF_PLAIN_BF16 / TF_PLAIN_FP16 / TF_SPLIT_BF16):
        what a fit's numbers have to be read against - fp32-equivalent by default, the plain 16-bit modes are opt-ins"""
        c = self.cfg
        if c.plain_bf16:
            return ("plain fp16 operands (NIC_FLAG_FP16), fp32 accumulate: outputs ~2e-5, gradients ~1e-3 from fp32" if getattr(c, "TF_PLAIN_FP16", False)
                    else "plain bf16 operands (NIC_FLAG_BF16), fp32 accumulate: outputs ~2e-4, gradients ~5e-3 from fp32")
        if c.TF_SPLIT_BF16:
            return "split-bf16 operands (hi + lo pairs, NIC_FLAG_SPLIT_BF16), fp32 accumulate: fp32-equivalent (5e-6)"
        return "fp32 matrix products"

    # ------------------------------------------------------------------ geometry helpers
    def _method(self) -> int:
        return self.cfg.COMPRESSION_METHOD if self.cfg.FP_DIMENSION == 3 else 1

    def _geometry(self, fl, mip_level, sample_number, num_crops, method=None, **kw) -> fused.PathGeometry:
        c = self.cfg
        D = c.FP_DIMENSION
        if kw.get("bf16") and c.TF_PLAIN_FP16 and (c.FEATURE_PYRAMID_CHANNELS, c.PE_CHANNELS) == (12, 6):
            kw["fp16"] = True                                      # the plain 16-bit products on IEEE half operands (NIC_FLAG_FP16 takes precedence in the library)
        return fused.PathGeometry(dim=D, method=method or self._method(), step_number=pow(2, mip_level - (fl + 1) * 2), mip_level=mip_level,
                                  extent=(sample_number,) * D, num_crops=num_crops, channels=c.FEATURE_PYRAMID_CHANNELS,
                                  pe_channels=c.PE_CHANNELS, hidden=c.HIDDEN_LAYER_CHANNELS, use_tri_pe=c.TF_USE_TRI_PE,
                                  num_bits=c.FP_BITS, **kw)

    def train_sample_number(self, mip_level: int) -> int:
        """2D hard-codes 2^max(0, 8 - mip) (image_compression.py:78); 3D uses CROP_MIP_LEVEL (:110,144)"""
        if self.cfg.FP_DIMENSION == 2:
            return pow(2, max(0, 8 - mip_level))
        return pow(2, max(0, self.cfg.CROP_MIP_LEVEL - mip_level))

    # ------------------------------------------------------------------ decoder input builders
    def create_decoder_input_2d(self, fp, coord, num_crops, fl, mip_level):
        """[N, Cin] for ``num_crops`` crops at ``coord`` (image_compression.py:71-100); differentiable w.r.t. fp"""
        geo = self._geometry(fl, mip_level, pow(2, max(0, 8 - mip_level)), num_crops)
        return fused.encode_differentiable(geo, fp[2 * fl], fp[2 * fl + 1], coord)

    def create_decoder_input_3d(self, fp, coord, num_crops, fl, mip_level, _method=3):
        """8 raw G0 corners + permuted-weight G1 blend + triangular PE (image_compression.py:103-134)"""
        geo = self._geometry(fl, mip_level, pow(2, max(0, self.cfg.CROP_MIP_LEVEL - mip_level)), num_crops, method=_method)
        return fused.encode_differentiable(geo, fp[2 * fl], fp[2 * fl + 1], coord)

    def create_decoder_input_3d_v2(self, fp, coord, num_crops, fl, mip_level):
        """4 tetrahedral G0 corners + G1 blend + sinusoidal PE (image_compression.py:137-167)"""
        return self.create_decoder_input_3d(fp, coord, num_crops, fl, mip_level, _method=4)

    def finally_decode_input_2d(self, fp, image_size, mip_level, x=0, y=0):
        """one tile of side ``image_size`` at origin (x, y) (image_compression.py:170-181)"""
        fl = self.feature_pyramid_mip_levels_dict[mip_level]
        return fused.encode(self._geometry(fl, mip_level, image_size, 1), fp[2 * fl], fp[2 * fl + 1], [[x, y]])

    def finally_decode_input_3d(self, fp, image_size, mip_level, x=0, y=0, z=0, _method=3):
        """image_compression.py:184-196"""
        fl = self.feature_pyramid_mip_levels_dict[mip_level]
        return fused.encode(self._geometry(fl, mip_level, image_size, 1, method=_method), fp[2 * fl], fp[2 * fl + 1], [[x, y, z]])

    def finally_decode_input_3d_v2(self, fp, image_size, mip_level, x=0, y=0, z=0):
        """image_compression.py:199-211"""
        return self.finally_decode_input_3d(fp, image_size, mip_level, x, y, z, _method=4)

    # ------------------------------------------------------------------ sampler (image_compression.py:26-50)
    def random_crop_dataset(self, datasets, crop_size, num_crops, uniform_distribution, dim=2):
        """LOD from python ``random``, ``num_crops`` origins from torch.randint on the CPU generator (one range for all
        axes), targets [num_crops, n, 3] sliced from the resident image; returns (targets, origins (host), lod)."""
        c = self.cfg
        if uniform_distribution:
            lod = random.randint(0, c.MAX_MIP_LEVEL)
        else:
            lod = min(int(math.floor(-math.log2(random.random()) / 2)), c.MAX_MIP_LEVEL)
        dataset = datasets[lod]
        data_size = dataset.shape[1]
        re_crop = max(1, crop_size // pow(2, lod))
        crops, coord = [], []
        for _ in range(num_crops):
            start = torch.randint(0, data_size - re_crop + 1, (dim,))
            sl = tuple(slice(int(start[d]), int(start[d]) + re_crop) for d in range(dim))
            crops.append(dataset[(slice(None), *sl)].reshape(3, -1).T)
            coord.append(start)
        return torch.stack(crops), torch.stack(coord), lod

    def random_crop_origins(self, datasets, crop_size, num_crops, uniform_distribution, dim=2):
        """the sampler without the crops: the SAME host RNG calls in the same order as :meth:`random_crop_dataset` (python
        ``random`` for the LOD, ``torch.randint`` per crop), so both draw identical (origins, lod) from identical seeds; the
        fused step then reads its targets straight from the resident image (``fused.TargetImage``)."""
        c = self.cfg
        if uniform_distribution:
            lod = random.randint(0, c.MAX_MIP_LEVEL)
        else:
            lod = min(int(math.floor(-math.log2(random.random()) / 2)), c.MAX_MIP_LEVEL)
        data_size = datasets[lod].shape[1]
        re_crop = max(1, crop_size // pow(2, lod))
        coord = [torch.randint(0, data_size - re_crop + 1, (dim,)) for _ in range(num_crops)]
        return torch.stack(coord), lod

    def set_images(self, images: Sequence[torch.Tensor], den: float = 255.0) -> None:
        """the resident dataset, one ``[3, S(, S), S]`` tensor per LOD (image_compression.py:429-477): fp32 like the reference's,
        or the uint8 codes they were made from (``ToTensor`` = u / 255; the 3D loader = u / 256) at a quarter of the memory"""
        self.images = [im.to(self.device) for im in images]
        self._targets = [fused.TargetImage(im, den) for im in self.images]
        if self.cfg.FP_DIMENSION == 2 and all(im.dtype == torch.uint8 and im.dim() == 3 for im in self.images):
            # 2D uint8 codes: the fused step reads its targets from an interleaved RGBX copy of every level (one dword per sample,
            # the 16-sample kernel's own target mode: 189 against 204 us per default step); same codes, same u / den
            from .sampler import rgbx_interleave
            self._targets = [fused.TargetImage(rgbx_interleave(im), den, rgbx=True) for im in self.images]
        self._sampler = None
        if self.cfg.TF_DEVICE_SAMPLER:
            # device-side sampler (sampler.py): RGBX levels built once from the level-0 codes, origins drawn on the device
            from .sampler import DeviceSampler, build_rgbx_pyramid
            if self.images[0].dtype != torch.uint8:
                raise ValueError("TF_DEVICE_SAMPLER needs the uint8 codes of the image (set_images([...uint8...]))")
            self._targets = build_rgbx_pyramid(self.images[0], self.cfg.MAX_MIP_LEVEL + 1, den, self.cfg.TF_MIP_FILTER)
            self._sampler = DeviceSampler(self.cfg.SAMPLER_SEED, self.device, self.cfg.NUM_CROPS)

    # ------------------------------------------------------------------ one training iteration
    def train_step(self, fp, epoch: int, fused_step: bool = True, noise_seed: int = 7):
        """body of the loop in train_models (image_compression.py:221-269); returns the loss (device scalar)"""
        c = self.cfg
        D = c.FP_DIMENSION
        resident = fused_step and len(getattr(self, "_targets", ())) == len(self.images) and len(self.images) > 0
        if fused_step and getattr(self, "_sampler", None) is not None:
            # device-side sampler: LOD on the host (pure function of seed and step), origins written on the device, targets from the RGBX level
            sizes = [t.spatial[0] for t in self._targets]
            coord, lod = self._sampler.draw(epoch, self._uniform(), c.MAX_MIP_LEVEL, c.CROP_SIZE, sizes, c.NUM_CROPS, D)   # coord stays on the device
            fl = self.feature_pyramid_mip_levels_dict[lod]
            resident = fp[2 * fl].requires_grad
            target = self._targets[lod] if resident else self._crops_rgbx(self._targets[lod], coord.cpu(), max(1, c.CROP_SIZE // pow(2, lod)))
        elif resident:
            coord, lod = self.random_crop_origins(self.images, c.CROP_SIZE, c.NUM_CROPS, self._uniform(), dim=D)
            fl = self.feature_pyramid_mip_levels_dict[lod]
            resident = fp[2 * fl].requires_grad
            target = self._targets[lod] if resident else self._crops(self.images[lod], coord, max(1, c.CROP_SIZE // pow(2, lod)))
        else:
            inputs, coord, lod = self.random_crop_dataset(self.images, c.CROP_SIZE, c.NUM_CROPS, self._uniform(), dim=D)
            fl = self.feature_pyramid_mip_levels_dict[lod]
            target = inputs.reshape(-1, 3)
        noisy = epoch < c.NUM_EPOCHS * 0.95
        plan_used = None
        if fused_step and fp[2 * fl].requires_grad and not getattr(self, "_no_fused_kernel", False):
            try:
                out, plan_used = self._fused_train_launch(fp, fl, lod, coord, target, noisy, noise_seed, epoch)
            except _lib.Unsupported:
                # a flag combination the fused kernels do not specialise (HIDDEN_LAYER_CHANNELS != 64, other channel counts, depths): the
                # layer-wise route below - nic_encode, the general decoder kernels, nic_encode_backward - serves it from now on
                self._no_fused_kernel = True
                if isinstance(target, fused.TargetImage):
                    target = self._materialise_target(target, coord, lod)
                return self._train_step_tail(fp, epoch, fl, lod, coord, target, noisy, noise_seed, None, layerwise=True)
            for t in fp:                                           # optimizer.zero_grad(set_to_none=True) without its hooks: the other levels must not step
                t.grad = None
            fp[2 * fl].grad, fp[2 * fl + 1].grad = out.grad_g0, out.grad_g1
            for p, g in zip(self.decoder.linear_params(), out.grad_mlp):
                p.grad = g
            loss = out.loss if out.loss.data_ptr() != out.flat.data_ptr() else out.loss.clone()   # a slot of the reused bucket is rewritten by the next step
            return self._train_step_tail(fp, epoch, fl, lod, coord, target, noisy, noise_seed, plan_used, loss=loss)
        if isinstance(target, fused.TargetImage):
            target = self._materialise_target(target, coord, lod)
        return self._train_step_tail(fp, epoch, fl, lod, coord, target, noisy, noise_seed, None, layerwise=getattr(self, "_no_fused_kernel", False))

    def _materialise_target(self, target, coord, lod):
        """[num_crops * n, 3] fp32 targets of the given origins from the resident image (the layer-wise route has no in-kernel target fetch)"""
        c = self.cfg
        re_crop = max(1, c.CROP_SIZE // pow(2, lod))
        coord_h = coord.cpu() if isinstance(coord, torch.Tensor) else coord
        if getattr(self, "_sampler", None) is not None:
            return self._crops_rgbx(target, coord_h, re_crop)
        return self._crops(self.images[lod], coord_h, re_crop)

    def _fused_train_launch(self, fp, fl, lod, coord, target, noisy, noise_seed, epoch):
        """the fused forward + backward launch of one step; returns (StepOutput, the StepPlan used or None)"""
        c = self.cfg
        plan_used = None
        geo = self._geometry(fl, lod, self.train_sample_number(lod), c.NUM_CROPS,
                             noise_mode=_lib.NIC_NOISE_KERNEL if noisy else _lib.NIC_NOISE_NONE,
                             noise_seed=noise_seed, noise_offset=epoch, split_bf16=bool(c.TF_SPLIT_BF16), bf16=c.plain_bf16)
        if isinstance(target, fused.TargetImage) and not os.environ.get("NIC_NO_PLAN"):
            # the steady state: one prepared launch plan (and one reused gradient bucket) per (level, LOD)
            plans = self.__dict__.setdefault("_plans", {})
            plan = plans.get((fl, lod))
            lin = self.decoder.linear_params()
            if plan is None or not plan.matches(fp[2 * fl], fp[2 * fl + 1], lin, target):
                plan = plans[(fl, lod)] = fused.StepPlan(geo, fp[2 * fl], fp[2 * fl + 1], lin, target)
            # the optimiser step of :266-269 rides on the reduction launch of the fused step (nic_path_desc.tail): two launches per step
            tail = self._step_tail(fp, fl, plan.gg0, plan.gg1, plan.gm)
            out = plan.run(coord, geo.noise_mode, noise_seed, epoch, tail=tail)
            plan_used = plan
        else:
            flats = self.__dict__.setdefault("_flat", {})      # one gradient bucket per level, reused: the optimiser's launch table stays valid
            out = fused.fused_forward_backward(geo, fp[2 * fl], fp[2 * fl + 1], coord, self.decoder.linear_params(), target, flat=flats.get(fl),
                                               tail=lambda gg0, gg1, gm: self._step_tail(fp, fl, gg0, gg1, gm))
            flats[fl] = out.flat
        return out, plan_used

    def _step_tail(self, fp, fl, gg0, gg1, gm):
        """``FusedAdam.step_tail`` for the level pair ``fl`` and the decoder on the given gradient buffers; None: the step stays a launch of its own
        (another optimiser class, NIC_NO_TAIL=1 - the A/B switch)"""
        if not isinstance(self.optimizer, FusedAdam) or os.environ.get("NIC_NO_TAIL") == "1":
            return None
        return self.optimizer.step_tail([(fp[2 * fl], gg0), (fp[2 * fl + 1], gg1)], list(zip(self.decoder.linear_params(), gm)))

    def _train_step_tail(self, fp, epoch, fl, lod, coord, target, noisy, noise_seed, plan_used, loss=None, layerwise=False):
        """the unfused forms of the step (when ``loss`` is None) and what follows every step: optimiser, scheduler, clamp"""
        c = self.cfg
        D = c.FP_DIMENSION
        if loss is None and c.DECODER_LINEAR_LAYERS != 3 and not layerwise and not getattr(self, "_no_fused_kernel", False):
            # deeper decoders: the tail after the freeze runs the fused op (forward kernel + recompute-backward kernel)
            geo = self._geometry(fl, lod, self.train_sample_number(lod), c.NUM_CROPS, split_bf16=bool(c.TF_SPLIT_BF16), bf16=c.plain_bf16,
                                 noise_mode=_lib.NIC_NOISE_KERNEL if noisy else _lib.NIC_NOISE_NONE, noise_seed=noise_seed, noise_offset=epoch)
            try:
                y = fused.fused_grid_mlp(geo, fp[2 * fl], fp[2 * fl + 1], coord, self.decoder.linear_params())
                loss = ((y - target) ** 2).mean()
                self.optimizer.zero_grad()
                loss.backward()
            except _lib.Unsupported:
                # a resumed run or a captured fit that reaches the tail without having met the refusal before (depth 2 / 4, other widths):
                # nothing has been launched; the layer-wise route below serves the rest of the fit
                self._no_fused_kernel = True
                loss = None
        if loss is not None:
            pass
        else:
            if D == 2:
                x = self.create_decoder_input_2d(fp, coord, c.NUM_CROPS, fl, lod)
            elif c.COMPRESSION_METHOD == 4:
                x = self.create_decoder_input_3d_v2(fp, coord, c.NUM_CROPS, fl, lod)
            else:
                x = self.create_decoder_input_3d(fp, coord, c.NUM_CROPS, fl, lod)
            if noisy:
                x = x + (torch.rand_like(x) - 0.5) / (2 ** c.FP_BITS)                                   # :250
            y = self.decoder(x)
            loss = ((y - target) ** 2).mean()                                                          # nn.MSELoss (:259)
            self.optimizer.zero_grad()
            loss.backward()
        self.optimizer.step()                             # Adam of both groups + the clamp of :269, one launch
        if plan_used is not None and isinstance(self.optimizer, FusedAdam):
            # .. which also zeroed the grid gradients of the bucket it read - when it really did (NIC_ADAM_ZERO_GRAD on both buffers);
            # otherwise the plan zeroes the bucket itself before its next launch
            plan_used.clean = self.optimizer.zeroed_in_last_step(plan_used.gg0, plan_used.gg1)
        self.scheduler.step()
        if not isinstance(self.optimizer, FusedAdam):
            fp_quantize_clamp(fp, fl, c.FP_BITS)                                                       # :269
        return loss.detach()

    def _crops(self, dataset: torch.Tensor, coord: torch.Tensor, re_crop: int) -> torch.Tensor:
        """[num_crops * n, 3] fp32 targets of the given origins, materialised (the unfused tail after the freeze)"""
        D = dataset.dim() - 1
        den = self._targets[0].den if dataset.dtype == torch.uint8 else 1.0
        rows = []
        for start in coord:
            sl = tuple(slice(int(start[d]), int(start[d]) + re_crop) for d in range(D))
            rows.append(dataset[(slice(None), *sl)].reshape(3, -1).T)
        t = torch.cat(rows)
        if dataset.dtype != torch.uint8:
            return t
        lut = (torch.arange(256, dtype=torch.float32) / den).to(t.device)    # divided on the host: correctly rounded, like ToTensor
        return lut[t.long()]

    def _crops_rgbx(self, timg, coord: torch.Tensor, re_crop: int) -> torch.Tensor:
        """[num_crops * n, 3] fp32 targets cut out of an RGBX level (the unfused tail after the freeze, device-sampler mode)"""
        w = timg.image
        D = w.dim()
        rows = []
        for start in coord:
            sl = tuple(slice(int(start[d]), int(start[d]) + re_crop) for d in range(D))
            rows.append(w[sl].reshape(-1))
        codes = torch.cat(rows)
        lut = (torch.arange(256, dtype=torch.float32) / timg.den).to(w.device)
        return torch.stack([lut[((codes >> (8 * ch)) & 255).long()] for ch in range(3)], dim=1)

    def _uniform(self) -> bool:
        self._acc = getattr(self, "_acc", 0.0) + self.cfg.UNIFORM_DISTRIBUTION_RATE                    # :221-226
        if self._acc >= 1.0:
            self._acc -= 1.0
            return True
        return False

    def train_models(self, fp, fused_step: bool = True, log_every: int = 0):
        """image_compression.py:215-303 without the tensorboard / file side effects.  NOTE: FusedAdam zeroes the grid gradients it has read
        (NIC_ADAM_ZERO_GRAD), so after a step ``grid.grad`` reads as zeros where torch.optim.Adam leaves the gradient in place.  After 95 % of the steps the grids are
        frozen and training continues on quantised copies (local rebinding, like the reference); losses stay on the
        device (no per-step host sync)."""
        c = self.cfg
        frozen = False
        for epoch in range(c.NUM_EPOCHS):
            if epoch > c.NUM_EPOCHS * 0.95 and not frozen:
                fp_freeze(fp)
                fp = fp_all_quantize(fp, c.FP_BITS)
                frozen = True
            loss = self.train_step(fp, epoch, fused_step)
            self.loss_history.append(loss)
            self.step_count += 1
            if log_every and (epoch + 1) % log_every == 0:
                print(f"Epoch [{epoch + 1}/{c.NUM_EPOCHS}], Loss: {loss.item():.4f}", flush=True)
        return fp

    def train_models_graph(self, fp, steps_per_graph: int = 8, log_every: int = 0, noise_seed: int = 7, time_replays: bool = False):
        """``train_models`` with the host out of the loop: the noisy, unfrozen part of the schedule (epochs below 0.95 NUM_EPOCHS,
        image_compression.py:227-254) runs as replays of ONE captured hipGraph of ``steps_per_graph`` steps - [draw the crop origins from the
        device step counter | fused forward + backward with the step number added to the noise offset | Adam + clamp with the step's learning
        rates from a device table filled once with the cosine schedule] - no argument is rewritten, no upload, no Python per step.  The
        reference's own launchers (320 000 steps of 8 x 32^3 samples) are host-bound everywhere else: their step needs ~0.1 ms of GPU.
        Same arithmetic as ``train_models`` with TF_DEVICE_SAMPLER (same origins, noise, learning rates: the losses agree to the order of the
        atomic sums).  Needs TF_DEVICE_SAMPLER (origins on the device) and TF_NO_MIP (the LOD fixes the launch geometry); the tail from
        0.95 NUM_EPOCHS on runs through ``train_step`` like ``train_models``."""
        c = self.cfg
        if getattr(self, "_sampler", None) is None:
            raise ValueError("train_models_graph draws its crop origins on the device: TF_DEVICE_SAMPLER=True and set_images([uint8 codes])")
        if c.MAX_MIP_LEVEL != 0:
            raise NotImplementedError("the LOD decides the launch geometry on the host: captured loops need TF_NO_MIP (the reference's default)")
        if not isinstance(self.optimizer, FusedAdam):
            raise TypeError("train_models_graph captures optim.FusedAdam's launch")
        D, N, e0 = c.FP_DIMENSION, c.NUM_EPOCHS, self.step_count
        e_end = min(N, int(math.ceil(N * 0.95)))                       # epochs e < 0.95 N: noise on, grids trainable
        n = max(0, e_end - e0)
        dev = self.device
        if n > 0:
            lod, fl = 0, self.feature_pyramid_mip_levels_dict[0]
            geo = self._geometry(fl, lod, self.train_sample_number(lod), c.NUM_CROPS, noise_mode=_lib.NIC_NOISE_KERNEL, noise_seed=noise_seed, noise_offset=0,
                                 split_bf16=bool(c.TF_SPLIT_BF16), bf16=c.plain_bf16)
            lin = self.decoder.linear_params()
            target = self._targets[lod]
            plans = self.__dict__.setdefault("_plans", {})
            plan = plans.get((fl, lod))
            if plan is None or not plan.matches(fp[2 * fl], fp[2 * fl + 1], lin, target):
                plan = plans[(fl, lod)] = fused.StepPlan(geo, fp[2 * fl], fp[2 * fl + 1], lin, target)
            for t in fp:
                t.grad = None
            fp[2 * fl].grad, fp[2 * fl + 1].grad = plan.gg0, plan.gg1
            for p_, g_ in zip(lin, plan.gm):
                p_.grad = g_
            pairs = [(fp[2 * fl], plan.gg0), (fp[2 * fl + 1], plan.gg1)] + list(zip(lin, plan.gm))
            adam = self.optimizer.dev_table(pairs, e0, self.scheduler.peek(n))
            if not (plan.gg0.data_ptr() in adam.zeroed and plan.gg1.data_ptr() in adam.zeroed):
                raise RuntimeError("the captured step relies on the optimiser zeroing the grid-gradient bucket (FusedAdam.zero_grad_in_step)")
            plan.flat.zero_()
            counters = torch.tensor([e0, e0], dtype=torch.int64, device=dev)
            step_ptr = counters.data_ptr() + 8
            hist = torch.zeros(e_end, dtype=torch.float32, device=dev)
            loss_slot = torch.zeros(1, dtype=torch.float32, device=dev)
            lib = _lib.load()
            ws = torch.empty(int(lib.nic_workspace_bytes(ctypes.byref(plan.d))), dtype=torch.uint8, device=dev)
            re_crop = max(1, c.CROP_SIZE // pow(2, lod))
            rng = int(target.spatial[0]) - re_crop + 1
            org = torch.zeros(c.NUM_CROPS * D, dtype=torch.int32, device=dev)
            seed = self._sampler.seed

            # the optimiser rides on the reduction launch of the fused step (nic_path_desc.tail with the device schedule): sampler + 2 launches per step
            tail_s = adam.tail_struct(2) if (plan.pad is None and os.environ.get("NIC_NO_TAIL") != "1") else None

            def one_step():
                _lib.check(lib.nic_sampler_step_begin(seed, _lib.ptr(counters), c.NUM_CROPS, D, rng, _lib.ptr(org), _lib.ptr(loss_slot), _lib.ptr(hist),
                                                      hist.numel(), _lib.stream_ptr(dev)), "nic_sampler_step_begin")
                plan.launch_dev(org, step_ptr, loss_slot, ws, _lib.NIC_NOISE_KERNEL, noise_seed, tail_struct=tail_s)
                if tail_s is None:
                    adam.launch(step_ptr)

            with torch.cuda.device(dev):
                one_step()                                              # the first step uncaptured: argument errors surface here, not inside a capture
                done = 1
                K = max(1, int(steps_per_graph))
                if (n - done) >= K:
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph):
                        for _ in range(K):
                            one_step()
                    if time_replays:                                    # benchmarks: wall time of the replay loop alone (two host syncs)
                        torch.cuda.synchronize(dev)
                        t0, d0 = time.perf_counter(), done
                    while n - done >= K:
                        graph.replay()
                        done += K
                        if log_every and done % log_every < K:
                            print(f"Epoch [{e0 + done}/{N}]", flush=True)
                    if time_replays:
                        torch.cuda.synchronize(dev)
                        self.graph_timing = {"steps": done - d0, "seconds": time.perf_counter() - t0, "steps_per_graph": K}
                while done < n:                                         # the remainder, same launches without a capture
                    one_step()
                    done += 1
                hist[e_end - 1:e_end].copy_(loss_slot)                  # the last step's loss has no successor to file it
            self._graph_keep = (counters, hist, loss_slot, ws, org, adam, tail_s)   # alive until the stream has drained
            adam.commit(n)
            self.scheduler.advance(n)
            plan.clean = True
            plan.steps += n
            self.loss_history.extend(hist[e] for e in range(e0, e_end))
            self.step_count += n
            for _ in range(n):
                self._uniform()                                         # the loop's uniform-LOD accumulator advances per step (unused at MAX_MIP_LEVEL 0)
        # the tail: no noise at 0.95 N exactly (Q4), then frozen grids / quantised copies (image_compression.py:227-231)
        frozen = False
        for epoch in range(self.step_count, N):
            if epoch > N * 0.95 and not frozen:
                fp_freeze(fp)
                fp = fp_all_quantize(fp, c.FP_BITS)
                frozen = True
            loss = self.train_step(fp, epoch, True, noise_seed)
            self.loss_history.append(loss)
            self.step_count += 1
        return fp

    # ------------------------------------------------------------------ decode (image_compression.py:307-346)
    def decode_image(self, fp, arc_decoder, mip_level, pr=False, div_size=10):
        """full image at ``mip_level`` as [S, S(, S), 3] (first axis = x): one fused encode+decoder launch per tile, tiles of
        side <= 2^div_size like the reference.  ``fp`` may be the fp32 grids or the STORED uint8 grids (``fp_savable`` /
        ``load_compressed``): those are dequantised inside the gather (``nic_fused_forward_u8``), bit-identical to ``fp_load``
        followed by the fp32 decode."""
        c = self.cfg
        D = c.FP_DIMENSION
        with torch.no_grad():
            power = c.MAX_MIP_LEVEL - mip_level
            div_slice = pow(2, max(power - div_size, 0))
            decode_size = c.IMAGE_SIZE // pow(2, mip_level)
            fl = self.feature_pyramid_mip_levels_dict[mip_level]
            params = arc_decoder.linear_params()
            stored = fp[2 * fl].dtype == torch.uint8
            if stored and len(params) != 6 and D != 2:      # 5-layer decoders decode from the stored codec in 2D; elsewhere dequantise once (fp_load), then decode
                from .fp_def import fp_load
                fp = fp_load(fp, c.FP_BITS, torch.float32)
                stored = False
            ga, gb = fp[2 * fl], fp[2 * fl + 1]
            # the decode runs the arithmetic the fit was trained in: plain-bf16 fits (TF_PLAIN_BF16) decode on the plain-bf16 forward kernels and gather
            # from the same 16-bit mirrors the training step read; split / fp32 fits on the split / fp32 inference kernels; fits that fell back to the
            # layer-wise fp32 route (no fused kernel for their widths / depth) decode layer-wise in fp32
            plain = bool(c.plain_bf16)
            if getattr(self, "_no_fused_kernel", False):
                self._no_fused_decode = True
            if D == 3 and not stored and not plain:
                ga, gb = ga.detach(), gb.detach()           # the 3D split / fp32 kernels read the fp32 masters (16-bit mirrors: plain-bf16 kernels, 2D split kernels)
            run_fused = (lambda geo, org: fused.fused_forward_u8(geo, ga, gb, org, params)) if stored else \
                        (lambda geo, org: fused.fused_forward(geo, ga, gb, org, params))

            def run(geo, org):
                if not getattr(self, "_no_fused_decode", False):
                    try:
                        return run_fused(geo, org)
                    except _lib.Unsupported:
                        self._no_fused_decode = True       # widths / depths without a fused kernel: encode + the general decoder kernels
                if stored:
                    from .fp_def import fp_load
                    a, b = fp_load([ga, gb], c.FP_BITS, torch.float32)
                else:
                    a, b = ga.detach(), gb.detach()
                return fused.DecoderFunction.apply(fused.encode(geo, a, b, org), *[t.detach() for t in params])
            split = bool(c.TF_SPLIT_BF16)                      # split-bf16 products (every layout's inference kernel; 2 x faster, outputs within 3e-7)
            # (channel counts other than the reference's defaults have fused kernels in the plain-bf16 family only: without TF_PLAIN_BF16 the fused
            #  entry points answer NIC_E_UNSUPPORTED for them and `run` decodes layer-wise in fp32 - like such a fit trains)
            wide = plain
            if div_slice == 1:
                y = run(self._geometry(fl, mip_level, decode_size, 1, split_bf16=split, bf16=wide), [[0] * D])
                return y.reshape(*([decode_size] * D), 3)
            if D != 2:
                raise NotImplementedError("tiled decode is 2D only, like the reference (image_compression.py:329-345)")
            s = decode_size // div_slice
            result = torch.zeros(decode_size, decode_size, 3, dtype=torch.float32, device=self.device)
            geo = self._geometry(fl, mip_level, s, 1, split_bf16=split, bf16=wide)
            for i in range(div_slice * div_slice):
                tx, ty = i % div_slice, i // div_slice
                result[s * tx:s * (tx + 1), s * ty:s * (ty + 1), :] = run(geo, [[s * tx, s * ty]]).reshape(s, s, 3)
            return result

    # ------------------------------------------------------------------ stored form (image_compression.py:370-397)
    def save_compressed(self, fp, fp_path: str, decoder_path: str) -> None:
        """the two files the reference writes: ``torch.save`` of the list of uint8 grids (``fp_savable``, :376,:383) and of the
        decoder's ``state_dict`` (:380) - readable by the reference and by :meth:`load_compressed`"""
        from .fp_def import fp_savable
        torch.save([g.cpu() for g in fp_savable(fp, self.cfg.FP_BITS, torch.uint8)], fp_path)
        torch.save({k: v.cpu() for k, v in self.decoder.state_dict().items()}, decoder_path)

    def load_compressed(self, fp_path: str, decoder_path: str, dequantise: bool = False):
        """reads files written by :meth:`save_compressed` or by the reference (:391-396).  Returns the grids as stored (uint8,
        on the device: decode them directly) or, with ``dequantise``, as fp32 like ``fp_load`` (fp_def.py:258-263)."""
        from .fp_def import fp_load
        self.decoder.load_state_dict(torch.load(decoder_path, map_location="cpu"))
        self.decoder.eval()
        stored = [g.to(self.device) for g in torch.load(fp_path, map_location="cpu")]
        if any(g.dtype != torch.uint8 for g in stored):
            raise NotImplementedError("only the reference's default 8-bit uint8 container is supported")
        return fp_load(stored, self.cfg.FP_BITS, torch.float32) if dequantise else stored

    def psnr(self, fp, mip_level: int = 0):
        """PSNR (peak 256) of the decode against the resident image (image_compression.py:283-289)"""
        D = self.cfg.FP_DIMENSION
        rec = self.decode_image(fp, self.decoder, mip_level)
        perm = (1, 2, 0) if D == 2 else (1, 2, 3, 0)
        ref = self.images[mip_level]
        if ref.dtype == torch.uint8:                       # resident codes (set_images): the value ToTensor / the 3D loader held
            ref = (torch.arange(256, dtype=torch.float32) / self._targets[mip_level].den).to(ref.device)[ref.long()]
        ref = ref.permute(*perm).contiguous()
        return calculate_psnr(quantize_to_bit(rec, self.cfg.OUTPUT_BITS), quantize_to_bit(ref, self.cfg.OUTPUT_BITS))
