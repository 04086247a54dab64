"""Multi-level grid mode (BASELINE.json config 2 names a "16-level grid"; SURVEY 8d config 2: "an extended mode with L mip levels ..
no oracle -> results pinned only through shared primitives").

The reference never reads more than ONE (G0, G1) level pair per sample: ``create_pyramid`` builds ``levels`` pairs, and a crop at mip
level m gathers from pair ``fl = clamp(m // 2 - 1, 0, levels - 1)`` only (fp_def.py:24-34, image_compression.py:76-79).  This module is
the extension: a sample gathers from the first L pairs AT ONCE and the decoder sees them CONCATENATED,

    x = [ enc_0 | enc_1 | .. | enc_{L-1} | lod ],   enc_l = [G0_l corners (4 C) | sum of the blended G1_l corners (C) | PE_l (2 P)]

where enc_l is exactly what the reference computes for pair l at ``step_number = 2^(mip - 2 (l + 1))`` (image_compression.py:79-100): the
G0 cell of pair l is 4^(l+1) pixels wide, its G1 cell twice that, the positional encoding is taken on the G1-cell coordinate of THAT
pair.  Cin = L (5 C + 2 P) + 1.  (Concatenated, not summed: a sum would make the pairs interchangeable up to the PE phase.)

TWO routes.  (a) FUSED (``fused=True``, the default wherever a kernel exists: ``fused.ml_is_fused`` - (levels, C, n_linear) in {(2,4,3), (3,4,3),
(5,4,3), (2,4,5), (3,4,5), (2,12,3), (3,12,3)} with P = 6, H = 64, i.e. what fits 160 KB of LDS beside the [64, Cin] weight image): ONE launch per step -
``nic_fused_ml_forward_backward``: every pair's gathers, the in-kernel noise, the decoder in plain-bf16 products, the loss, the backward pass and one
gradient flush per touched cell and pair (csrc/fused_q16.hpp::QML) - then ``nic_adam_multi``.  (b) LAYER-WISE, for every other combination and
for the fp32 reference decode: every enc_l from ``nic_encode`` / ``nic_encode_backward`` (the kernels behind ``create_decoder_input_2d``), the
decoder from the general layer-wise kernels (``nic_decoder_general_*``) and the optimiser from ``nic_adam_multi``; the [N, Cin] input, its noise
and the loss are torch tensors there like in the reference (image_compression.py:248-259).  The parity tests hold (b) to the composition of the oracle's
``create_decoder_input`` and ``mlp_forward`` and (a) to the same composition with the oracle's bf16-emulating decoder.  Grids per pair are sized ``ceil(S / cell) + 1`` nodes per axis (the reference's
``base // 2^i + 1`` on its power-of-two squares), so non-square and non-power-of-two images (3840 x 2160) stay in bounds at every level.
Hashed indexing is not built: the reference's grids are dense and so are these (a 4K pyramid of 5 pairs is 8.3 M parameters,
7.8 M of them in pair 0).  2D only (config 2 is an image fit).
"""
from __future__ import annotations

import os

from typing import List, Optional, Sequence, Tuple, Union

import torch

from . import _lib, fused
from .image_compression import ColorDecoder
from .optim import CosineAnnealing, FusedAdam


def level_nodes(image_size: Sequence[int], level: int) -> Tuple[Tuple[int, ...], Tuple[int, ...]]:
    """nodes per axis (x, y) of G0 and G1 of pair ``level`` for an image of ``image_size`` = (S_x, S_y) samples at mip 0"""
    c0 = 4 ** (level + 1)
    g0 = tuple(-(-int(s) // c0) + 1 for s in image_size)
    g1 = tuple(-(-int(s) // (2 * c0)) + 1 for s in image_size)
    return g0, g1


def max_levels(image_size: Sequence[int]) -> int:
    """the reference's pair count for FEATURE_PYRAMID_SIZE = S // 4: (floor(log2(base)) + 1) // 2 (fp_def.py:8-20), on the shorter axis"""
    from .fp_def import return_pyramid_levels
    return max(return_pyramid_levels(max(min(int(v) for v in image_size) // 4, 1)), 1)


class MultiLevelField:
    """L level pairs + one decoder over their concatenated encodings (module docstring).  ``image_size``: (S_x, S_y), x = the image tensor's
    first spatial axis like everywhere in the reference (fp_def.py:81-86)."""

    def __init__(self, image_size: Union[int, Sequence[int]], levels: int, channels: int = 12, pe_channels: int = 6, hidden: int = 64,
                 n_linear: int = 3, num_bits: int = 8, device=None, use_tri_pe: bool = True, seed: Optional[int] = None, fused_step: Optional[bool] = None,
                 noise_seed: int = 7):
        self.image_size = (int(image_size),) * 2 if isinstance(image_size, int) else tuple(int(v) for v in image_size)
        if len(self.image_size) != 2:
            raise NotImplementedError("the multi-level mode is 2D")
        if not 1 <= levels <= max_levels(self.image_size):
            raise ValueError(f"1 .. {max_levels(self.image_size)} level pairs for an image of {self.image_size}")
        self.device = torch.device(device if device is not None else "cuda")
        if self.device.type != "cuda":
            raise RuntimeError("MultiLevelField needs a HIP device: there is no CPU implementation of this path")
        self.levels, self.channels, self.pe_channels, self.num_bits, self.use_tri_pe = levels, channels, pe_channels, num_bits, use_tri_pe
        if seed is not None:
            torch.manual_seed(seed)
        lo, hi = -(2 ** num_bits - 1) / 2 ** (num_bits + 1), 0.5                     # fp_def.py:42-43
        self.fp: List[torch.Tensor] = []
        for l in range(levels):
            for nodes in level_nodes(self.image_size, l):
                g = (hi - lo) * torch.rand(channels, nodes[1], nodes[0], device=self.device, dtype=torch.float32) + lo   # [C, Y, X], the reference's init expression
                self.fp.append(g.requires_grad_(True))
        self.per_level = 5 * channels + 2 * pe_channels
        self.cin = levels * self.per_level + 1
        self.decoder = ColorDecoder(self.cin, hidden, n_linear).to(self.device)
        self.optimizer = FusedAdam([{"params": self.fp, "lr": 0.01}, {"params": self.decoder.parameters(), "lr": 0.005}])   # image_compression.py:361-364
        self.optimizer.set_clamp(self.fp, lo, hi)                                      # fp_def.py:227-232, every pair is used by every step
        self.scheduler = None
        can = fused.ml_is_fused(levels, channels, pe_channels, hidden, n_linear)
        if fused_step and not can:
            raise _lib.Unsupported(f"no fused multi-level kernel for levels={levels}, channels={channels}, pe_channels={pe_channels}, hidden={hidden}, n_linear={n_linear}")
        self.fused_step = can if fused_step is None else bool(fused_step)
        self.noise_seed, self.steps = int(noise_seed), 0
        self._grads = None                                                             # fused route: one gradient tensor per grid / decoder parameter, reused
        self._loss = None

    def set_schedule(self, num_epochs: int) -> None:
        self.scheduler = CosineAnnealing(self.optimizer, T_max=num_epochs, eta_min=0)  # image_compression.py:365

    def geometry(self, level: int, extent: Sequence[int], num_crops: int, mip_level: int = 0) -> fused.PathGeometry:
        return fused.PathGeometry(dim=2, method=1, step_number=pow(2, mip_level - 2 * (level + 1)), mip_level=mip_level, extent=tuple(int(e) for e in extent),
                                  num_crops=num_crops, channels=self.channels, pe_channels=self.pe_channels, hidden=self.decoder.decoder[0].out_features,
                                  use_tri_pe=self.use_tri_pe, num_bits=self.num_bits)

    def decoder_input(self, coord, extent: Sequence[int], mip_level: int = 0, fp: Optional[List[torch.Tensor]] = None) -> torch.Tensor:
        """[N, Cin] for the crops at ``coord`` ([num_crops, 2] origins), differentiable w.r.t. every grid"""
        fp = self.fp if fp is None else fp
        num_crops = len(coord)
        cols = []
        for l in range(self.levels):
            e = fused.encode_differentiable(self.geometry(l, extent, num_crops, mip_level), fp[2 * l], fp[2 * l + 1], coord)
            cols.append(e if l == self.levels - 1 else e[:, :-1])                      # the LOD column once, at the end
        return torch.cat(cols, dim=1) if len(cols) > 1 else cols[0]

    def forward(self, coord, extent: Sequence[int], mip_level: int = 0, noise: bool = False) -> torch.Tensor:
        x = self.decoder_input(coord, extent, mip_level)
        if noise:
            x = x + (torch.rand_like(x) - 0.5) / (2 ** self.num_bits)                 # image_compression.py:250
        return self.decoder(x)

    def train_step(self, coord, extent: Sequence[int], target: torch.Tensor, noise: bool = True, accumulate: bool = False, scale: float = 1.0,
                   step: bool = True) -> torch.Tensor:
        """one step on the crops at ``coord`` with targets [N, 3] (image_compression.py:239-269).  ``accumulate`` / ``scale`` / ``step``: a
        whole-image pass walked in chunks - gradients add up over the chunks (each chunk's MSE scaled by its share), one optimiser step at the end"""
        if self.fused_step:
            return self._fused_train_step(coord, extent, target, noise, accumulate, scale, step)
        if not accumulate:
            self.optimizer.zero_grad()
        y = self.forward(coord, extent, noise=noise)
        loss = ((y - target) ** 2).mean() * scale                                      # nn.MSELoss (:259)
        loss.backward()
        if step:
            self.optimizer.step()                                                      # Adam of both groups + the clamp of :269, one launch
            if self.scheduler is not None:
                self.scheduler.step()
        return loss.detach()

    def _fused_train_step(self, coord, extent, target, noise, accumulate, scale, step) -> torch.Tensor:
        """the step as ONE fused launch + the optimiser launch.  Gradient tensors persist (the optimiser zeroes the grids' after reading them -
        FusedAdam.zero_grad_in_step - so a step needs no fill launch); with ``accumulate`` the grid gradients add up in place over the chunks of a
        pass, the decoder's (overwritten per launch by the fixed-order reduction) are added on the side."""
        num_crops = len(coord)
        params = self.decoder.linear_params()
        if self._grads is None:
            self._grads = ([torch.zeros_like(g) for g in self.fp], [torch.zeros_like(p) for p in params], [torch.zeros_like(p) for p in params])
            self._loss = torch.zeros(1, dtype=torch.float32, device=self.device)
            for t, g in zip(self.fp, self._grads[0]):
                t.grad = g
            for t, g in zip(params, self._grads[1]):
                t.grad = g
            self.optimizer.zero_grad_in_step(self.fp)
        gfp, gmlp, gtmp = self._grads
        n = num_crops * int(extent[0]) * int(extent[1])
        geo = fused.PathGeometry(dim=2, method=1, step_number=0.25, mip_level=0, extent=tuple(int(e) for e in extent), num_crops=num_crops, channels=self.channels,
                                 pe_channels=self.pe_channels, hidden=64, use_tri_pe=self.use_tri_pe, num_bits=self.num_bits,
                                 noise_mode=_lib.NIC_NOISE_KERNEL if noise else _lib.NIC_NOISE_NONE, noise_seed=self.noise_seed, noise_offset=self.steps,
                                 loss_scale=float(scale) / (3.0 * n))
        if not accumulate and not getattr(self, "_grid_grads_clean", False):
            for g in gfp:
                g.zero_()
        tail = None
        if step and not accumulate and isinstance(self.optimizer, FusedAdam) and os.environ.get("NIC_NO_TAIL") != "1":
            # a whole step in this launch: the optimiser rides on the reduction of the decoder records (nic_path_desc.tail) - two launches per step
            tail = self.optimizer.step_tail(list(zip(self.fp, gfp)), list(zip(params, gmlp)))
        out = fused.fused_ml_forward_backward(geo, self.fp, coord, params, target, grads=gfp, mlp_grads=gtmp if accumulate else gmlp, loss=self._loss, tail=tail)
        if accumulate:
            for a, b in zip(gmlp, gtmp):
                a.add_(b)
        self._grid_grads_clean = False
        if step:
            self.optimizer.step()                                                      # Adam of both groups + the clamp of :269 + zeroing of the grid gradients, one launch
            self._grid_grads_clean = isinstance(self.optimizer, FusedAdam) and self.optimizer.zeroed_in_last_step(*gfp)
            if self.scheduler is not None:
                self.scheduler.step()
            self.steps += 1
        return out.loss.detach().clone()                                              # this call's (scaled) loss, like the layer-wise route

    @torch.no_grad()
    def decode(self, tile: int = 1024, fused_forward: bool = False) -> torch.Tensor:
        """the whole image [S_x, S_y, 3], tiles of side <= ``tile`` like decode_image (image_compression.py:307-346): layer-wise in fp32 (the
        reference arithmetic), or - ``fused_forward`` - one fused launch per tile in the plain-bf16 products the fused step trains in"""
        sx, sy = self.image_size
        out = torch.empty(sx, sy, 3, dtype=torch.float32, device=self.device)
        fp = [g.detach() for g in self.fp]
        params = [p.detach() for p in self.decoder.linear_params()]
        for x0 in range(0, sx, tile):
            for y0 in range(0, sy, tile):
                ext = (min(tile, sx - x0), min(tile, sy - y0))
                if fused_forward:
                    geo = fused.PathGeometry(dim=2, method=1, step_number=0.25, mip_level=0, extent=ext, num_crops=1, channels=self.channels,
                                             pe_channels=self.pe_channels, hidden=64, use_tri_pe=self.use_tri_pe, num_bits=self.num_bits)
                    out[x0:x0 + ext[0], y0:y0 + ext[1]] = fused.fused_ml_forward(geo, fp, [[x0, y0]], params).reshape(ext[0], ext[1], 3)
                    continue
                x = self.decoder_input([[x0, y0]], ext, fp=fp)
                out[x0:x0 + ext[0], y0:y0 + ext[1]] = fused.DecoderFunction.apply(x, *params).reshape(ext[0], ext[1], 3)
        return out
