"""The optimiser step of the training loop in ONE launch.

The reference registers every grid and the decoder in one ``torch.optim.Adam`` (two parameter groups, lr 0.01 / 0.005), steps a
``CosineAnnealingLR`` after it and then clamps the two grids of the level just used (image_compression.py:266-269, 361-365;
fp_def.py:227-232).  :class:`FusedAdam` is a ``torch.optim.Optimizer`` with ``Adam``'s state layout (``step``, ``exp_avg``,
``exp_avg_sq`` per parameter, so ``state_dict()`` round-trips with torch's) whose ``step()`` is a single ``nic_adam_multi``
launch over every parameter that has a gradient; parameters registered through :meth:`set_clamp` are clamped in the same pass,
which makes the reference's ``fp_quantize_clamp`` call that follows an idempotent no-op.  Stock ``lr_scheduler`` classes work
on it unchanged (they only touch ``group['lr']``).
"""
from __future__ import annotations

import ctypes
from typing import Dict, Iterable, Optional, Tuple

import torch

from . import _lib


class FusedAdam(torch.optim.Optimizer):
    """``torch.optim.Adam`` (defaults: no weight decay, no amsgrad, not maximize) on libnicv2_hip's ``nic_adam_multi``."""

    def __init__(self, params, lr: float = 1e-3, betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._clamp: Dict[int, Tuple[float, float]] = {}
        self._mirror: Dict[int, torch.Tensor] = {}

    def set_clamp(self, params: Iterable[torch.Tensor], lo: float, hi: float) -> None:
        """clamp these parameters to [lo, hi] right after their update (fp_quantize_clamp, fp_def.py:227-232)"""
        for p in params:
            self._clamp[id(p)] = (float(lo), float(hi))

    def set_mirror(self, param: torch.Tensor, mirror16: torch.Tensor) -> None:
        """16-bit grid storage: ``param`` is the fp32 master the optimiser updates, ``mirror16`` a bfloat16 / float16 tensor of the same
        shape that the same launch rewrites with the rounded new values (what the fused kernels gather from, NIC_FLAG_GRID_*)."""
        if mirror16.dtype not in (torch.bfloat16, torch.float16) or mirror16.shape != param.shape or not mirror16.is_contiguous():
            raise ValueError("the mirror is a contiguous bfloat16 / float16 tensor of the parameter's shape")
        self._mirror[id(param)] = mirror16

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        # groups that share (betas, eps) - the reference's two groups do - go into the same launch; lr is per tensor
        batches: Dict[Tuple[float, float, float], list] = {}
        keep = []                                                # contiguous gradient copies must outlive the launch
        device: Optional[torch.device] = None
        for group in self.param_groups:
            b1, b2 = group["betas"]
            entries = batches.setdefault((float(b1), float(b2), float(group["eps"])), [])
            for p in group["params"]:
                if p.grad is None:
                    continue                                     # torch skips it and does not advance its step
                _lib.require_cuda_f32(p, "parameter")
                if not p.is_contiguous():
                    raise RuntimeError("FusedAdam updates parameters in place: they must be contiguous")
                if device is not None and p.device != device:
                    raise RuntimeError("FusedAdam: all parameters must live on one device")
                g = _lib.require_cuda_f32(p.grad, "gradient")
                keep.append(g)
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)               # host scalar, like torch's default (capturable=False)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                lo, hi = self._clamp.get(id(p), (1.0, -1.0))
                mir = self._mirror.get(id(p))
                entries.append(_lib.NicAdamTensor(p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                                                  p.numel(), int(st["step"].item()), float(group["lr"]), lo, hi,
                                                  0 if mir is None else mir.data_ptr(), 0 if mir is None else (1 if mir.dtype == torch.bfloat16 else 2), 0))
                device = p.device
        for (b1, b2, eps), entries in batches.items():
            for i in range(0, len(entries), _lib.NIC_ADAM_MAX_TENSORS):
                chunk = entries[i:i + _lib.NIC_ADAM_MAX_TENSORS]
                arr = (_lib.NicAdamTensor * len(chunk))(*chunk)
                with torch.cuda.device(device):
                    _lib.check(lib.nic_adam_multi(arr, len(chunk), b1, b2, eps, _lib.stream_ptr(device)), "nic_adam_multi")
        return loss
