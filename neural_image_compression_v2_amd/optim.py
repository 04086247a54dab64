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
        self._zero: set = set()

    def load_state_dict(self, state_dict) -> None:
        self._cache = self._tail_cache = None                    # the state tensors are replaced
        super().load_state_dict(state_dict)

    def add_param_group(self, param_group) -> None:
        self._cache = self._tail_cache = None
        super().add_param_group(param_group)

    def zero_grad_in_step(self, params: Iterable[torch.Tensor]) -> None:
        """the launch that updates these parameters also zeroes their gradient buffers (``NIC_ADAM_ZERO_GRAD``): the bucket a fused step
        accumulates into with atomics is clean for the next step without a fill kernel (``fused.StepPlan.clean``)"""
        for p in params:
            self._zero.add(id(p))
        self._cache = None

    def zeroed_in_last_step(self, *grads: torch.Tensor) -> bool:
        """True when the last ``step()`` launched every one of these gradient buffers with ``NIC_ADAM_ZERO_GRAD`` - i.e. they are known to be
        zero now.  ``fused.StepPlan.clean`` is set from this, never assumed: an optimiser rebuilt without ``zero_grad_in_step``, or a
        parameter outside its groups, leaves the plan to zero its bucket itself."""
        z = getattr(self, "_zeroed", frozenset())
        return all(g is not None and g.data_ptr() in z for g in grads)

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

    # ---- the step as the TAIL of the fused training launch (nic_path_desc.tail, csrc/nic_adam.hpp): the reduction of the decoder-gradient records
    # and this optimiser's update in ONE launch - the grids are streamed while the reduction walks the records, the decoder's parameters are updated
    # by the threads that finish their gradients.  Same arithmetic as step(), bit for bit.
    @torch.no_grad()
    def step_tail(self, stream_pairs, decoder_pairs) -> Optional["StepTail"]:
        """``stream_pairs``: [(parameter, its gradient buffer)] whose gradients are complete when the fused kernel ends (the grids);
        ``decoder_pairs``: the decoder's tensors with the buffers the step's reduction writes (``StepPlan.gm`` / ``StepOutput.grad_mlp``).  Returns
        the handle to pass as ``tail=`` to ``fused.fused_forward_backward`` / ``StepPlan.run`` - which commits it once the launch is queued - or
        None when this optimiser cannot ride on one launch (more than NIC_ADAM_MAX_TENSORS tensors, groups with different betas / eps): the
        caller then steps the ordinary way.  After a committed tail the next ``step()`` - the reference's call (image_compression.py:266) -
        launches nothing."""
        pairs = list(stream_pairs) + list(decoder_pairs)
        keys = {(float(g["betas"][0]), float(g["betas"][1]), float(g["eps"])) for g in self.param_groups}
        if len(keys) != 1 or not pairs or len(pairs) > _lib.NIC_ADAM_MAX_TENSORS:
            return None
        ptrs = tuple((p.data_ptr(), g.data_ptr()) for p, g in pairs)
        c = getattr(self, "_tail_cache", None)
        if c is None or c.ptrs != ptrs or c.n_stream != len(stream_pairs) or c.clamp != self._clamp or c.mirror != {k: id(v) for k, v in self._mirror.items()}:
            gidx = {id(p): gi for gi, g in enumerate(self.param_groups) for p in g["params"]}
            entries, zeroed = [], set()
            for i, (p, g) in enumerate(pairs):
                if id(p) not in gidx:
                    raise ValueError("a tail parameter is not registered in this optimiser")
                _lib.require_cuda_f32(p, "parameter")
                if not (p.is_contiguous() and g.is_contiguous() and g.dtype == torch.float32 and g.is_cuda and g.numel() == p.numel()):
                    raise RuntimeError("tail parameters and gradient buffers are contiguous fp32 device tensors of one size")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                lo, hi = self._clamp.get(id(p), (1.0, -1.0))
                mir = self._mirror.get(id(p))
                zero = id(p) in self._zero and i < len(stream_pairs)
                if zero:
                    zeroed.add(g.data_ptr())
                entries.append(_lib.NicAdamTensor(p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(),
                                                  int(st["step"].item()), 0.0, lo, hi, 0 if mir is None else mir.data_ptr(),
                                                  0 if mir is None else (1 if mir.dtype == torch.bfloat16 else 2), _lib.NIC_ADAM_ZERO_GRAD if zero else 0))
            (b1, b2, eps), = keys
            c = self._tail_cache = StepTail(self, (_lib.NicAdamTensor * len(entries))(*entries), len(stream_pairs), (b1, b2, eps), [p for p, _ in pairs],
                                            [gidx[id(p)] for p, _ in pairs], frozenset(zeroed), ptrs, dict(self._clamp), {k: id(v) for k, v in self._mirror.items()})
        for i, (p, gi) in enumerate(zip(c.params, c.gidx)):
            c.arr[i].step = int(self.state[p]["step"].item()) + 1
            c.arr[i].lr = float(self.param_groups[gi]["lr"])
        return c

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if getattr(self, "_tail_done", False):
            self._tail_done = False                              # this step's update rode on the fused launch (step_tail)
            return loss
        lib = _lib.load()
        if self._fast_step(lib):
            return loss
        # groups that share (betas, eps) - the reference's two groups do - go into the same launch; lr is per tensor
        batches: Dict[Tuple[float, float, float], list] = {}
        zeroed = set()
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
                zero_here = id(p) in self._zero and g.data_ptr() == p.grad.data_ptr()   # a contiguous COPY of the gradient is not the buffer to zero
                if zero_here:
                    zeroed.add(g.data_ptr())
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
                                                  0 if mir is None else mir.data_ptr(), 0 if mir is None else (1 if mir.dtype == torch.bfloat16 else 2),
                                                  _lib.NIC_ADAM_ZERO_GRAD if zero_here else 0))
                if id(p) in self._zero and not zero_here:
                    p.grad.zero_()                               # the promise holds for the caller's own buffer too
                device = p.device
        for (b1, b2, eps), entries in batches.items():
            for i in range(0, len(entries), _lib.NIC_ADAM_MAX_TENSORS):
                chunk = entries[i:i + _lib.NIC_ADAM_MAX_TENSORS]
                arr = (_lib.NicAdamTensor * len(chunk))(*chunk)
                with torch.cuda.device(device):
                    _lib.check(lib.nic_adam_multi(arr, len(chunk), b1, b2, eps, _lib.stream_ptr(device)), "nic_adam_multi")
        self._zeroed = frozenset(zeroed)
        self._remember(batches, device)
        return loss

    # ---- hipGraph-captured loops: the launch reads its per-step scalars from a device table (nic_adam_multi_dev)
    def dev_table(self, pairs, first_step_row: int, lrs_per_step) -> "DevAdam":
        """``pairs``: [(parameter, its gradient buffer)] - every parameter of the captured step.  ``lrs_per_step``: per future step a list of
        per-group learning rates (``CosineAnnealing.peek``).  Row ``first_step_row + k`` of the schedule holds the scalars of the k-th step
        from now: lr / bias_correction1 per group (formed in double, cast once - what nic_adam_multi does per call) and
        sqrt(bias_correction2).  All parameters must share one step count (they do when every step updates all of them) and one
        (betas, eps); at most two parameter groups (the reference's: grids, decoder)."""
        import math
        if len(self.param_groups) > 2:
            raise NotImplementedError("two parameter groups (grids, decoder), like the reference's optimiser (image_compression.py:361-364)")
        keys = {(float(g["betas"][0]), float(g["betas"][1]), float(g["eps"])) for g in self.param_groups}
        if len(keys) != 1:
            raise NotImplementedError("one (betas, eps) for all groups")
        (b1, b2, eps), = keys
        gidx = {id(p): gi for gi, g in enumerate(self.param_groups) for p in g["params"]}
        entries, steps, zeroed = [], set(), set()
        for p, g in pairs:
            if id(p) not in gidx:
                raise ValueError("a captured parameter is not registered in this optimiser")
            _lib.require_cuda_f32(p, "parameter")
            if not (p.is_contiguous() and g.is_contiguous() and g.dtype == torch.float32):
                raise RuntimeError("captured parameters and gradient buffers are contiguous fp32")
            st = self.state[p]
            if len(st) == 0:
                st["step"] = torch.tensor(0.0)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            steps.add(int(st["step"].item()))
            lo, hi = self._clamp.get(id(p), (1.0, -1.0))
            mir = self._mirror.get(id(p))
            flags = (_lib.NIC_ADAM_ZERO_GRAD if id(p) in self._zero else 0) | (_lib.NIC_ADAM_SCHED_COL1 if gidx[id(p)] == 1 else 0)
            if id(p) in self._zero:
                zeroed.add(g.data_ptr())
            entries.append(_lib.NicAdamTensor(p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(), 0, 0.0, lo, hi,
                                              0 if mir is None else mir.data_ptr(), 0 if mir is None else (1 if mir.dtype == torch.bfloat16 else 2), flags))
        if len(steps) != 1:
            raise RuntimeError("captured parameters must share one Adam step count")
        if len(entries) > _lib.NIC_ADAM_MAX_TENSORS:
            raise NotImplementedError(f"at most {_lib.NIC_ADAM_MAX_TENSORS} tensors per captured optimiser launch")
        s0 = steps.pop()
        n = len(lrs_per_step)
        import numpy as np
        rows = np.zeros((first_step_row + n, 4), dtype=np.float32)
        for k, lrs in enumerate(lrs_per_step):
            step = s0 + k + 1
            bc1, bc2 = 1.0 - math.pow(b1, float(step)), 1.0 - math.pow(b2, float(step))
            rows[first_step_row + k, 0] = np.float32(float(lrs[0]) / bc1)
            rows[first_step_row + k, 1] = np.float32(float(lrs[-1]) / bc1)
            rows[first_step_row + k, 2] = np.float32(math.sqrt(bc2))
        dev = pairs[0][0].device
        self._cache = None
        return DevAdam(self, (_lib.NicAdamTensor * len(entries))(*entries), len(entries), (b1, b2, eps), torch.from_numpy(rows).to(dev), [p for p, _ in pairs],
                       frozenset(zeroed), dev)

    # ---- the steady state of a training loop: the same parameters, the same gradient / state buffers (the fused step writes its gradients
    # into one reused bucket), one launch.  The table of the previous step is reused when every pointer in it is still current; only
    # the step counts and learning rates are rewritten (the full path above costs ~60 us of Python per step).
    def _remember(self, batches, device) -> None:
        self._cache = None
        if len(batches) != 1 or device is None:
            return
        (key, entries), = batches.items()
        if not entries or len(entries) > _lib.NIC_ADAM_MAX_TENSORS:
            return
        plist = [p for g in self.param_groups for p in g["params"] if p.grad is not None]
        if len(plist) != len(entries):
            return
        gidx = [gi for gi, g in enumerate(self.param_groups) for p in g["params"] if p.grad is not None]
        arr = (_lib.NicAdamTensor * len(entries))(*entries)
        self._cache = dict(key=key, arr=arr, params=plist, gidx=gidx, device=device, nparam=sum(len(g["params"]) for g in self.param_groups),
                           steps=[self.state[p]["step"] for p in plist], clamp=dict(self._clamp), mirror={k: id(v) for k, v in self._mirror.items()})

    def _fast_step(self, lib) -> bool:
        c = getattr(self, "_cache", None)
        if c is None:
            return False
        arr, plist = c["arr"], c["params"]
        if c["nparam"] != sum(len(g["params"]) for g in self.param_groups) or c["clamp"] != self._clamp \
                or c["mirror"] != {k: id(v) for k, v in self._mirror.items()}:
            self._cache = None
            return False
        n = 0
        for g in self.param_groups:
            if (float(g["betas"][0]), float(g["betas"][1]), float(g["eps"])) != c["key"]:
                self._cache = None
                return False
            for p in g["params"]:
                gr = p.grad
                if gr is None:
                    continue
                if n >= len(plist) or p is not plist[n] or gr.data_ptr() != arr[n].grad or p.data_ptr() != arr[n].param or gr.dtype != torch.float32 \
                        or not gr.is_contiguous():
                    self._cache = None
                    return False
                n += 1
        if n != len(plist):
            self._cache = None
            return False
        torch._foreach_add_(c["steps"], 1)
        for i, gi in enumerate(c["gidx"]):
            arr[i].step += 1
            arr[i].lr = float(self.param_groups[gi]["lr"])
        b1, b2, eps = c["key"]
        with torch.cuda.device(c["device"]):
            _lib.check(lib.nic_adam_multi(arr, len(plist), b1, b2, eps, _lib.stream_ptr(c["device"])), "nic_adam_multi")
        return True


class StepTail:
    """``FusedAdam.step_tail``'s handle: the nic_step_tail struct of one training step.  ``struct_ptr`` goes into nic_path_desc.tail; ``commit()``
    (called by the fused wrappers once the launch is queued) advances the optimiser's step counts and makes its next ``step()`` a no-op."""

    def __init__(self, opt, arr, n_stream, key, params, gidx, zeroed, ptrs, clamp, mirror):
        self.opt, self.arr, self.n_stream, self.key, self.params, self.gidx, self.zeroed = opt, arr, n_stream, key, params, gidx, zeroed
        self.ptrs, self.clamp, self.mirror = ptrs, clamp, mirror
        self.struct = _lib.NicStepTail()
        self.struct.tensors = ctypes.cast(arr, ctypes.c_void_p).value
        self.struct.count, self.struct.n_stream = len(arr), n_stream
        self.struct.beta1, self.struct.beta2, self.struct.eps = key
        self.steps = [opt.state[p]["step"] for p in params]

    @property
    def struct_ptr(self) -> int:
        return ctypes.addressof(self.struct)

    def decoder_grad_ptrs(self):
        return [self.arr[i].grad for i in range(self.n_stream, len(self.arr))]

    def commit(self) -> None:
        torch._foreach_add_(self.steps, 1)
        self.opt._zeroed = self.zeroed
        self.opt._tail_done = True
        self.opt._cache = None                                   # step()'s table holds the old step counts


class DevAdam:
    """the captured optimiser launch of ``FusedAdam.dev_table``: ``launch(step_dev_ptr)`` inside the capture, ``commit(n)`` once the replays
    are done (host-side step counts of the ``state_dict``)"""

    def __init__(self, opt, arr, count, key, sched, params, zeroed, device):
        self.opt, self.arr, self.count, self.key, self.sched, self.params, self.zeroed, self.device = opt, arr, count, key, sched, params, zeroed, device

    def tail_struct(self, n_stream: int) -> "_lib.NicStepTail":
        """this table as the tail of the captured fused step (nic_path_desc.tail with the device schedule: nic_fused_forward_backward_img_dev reads
        row *step_dev): the first ``n_stream`` entries are streamed (the grids), the rest are the decoder's tensors"""
        t = _lib.NicStepTail()
        t.tensors = ctypes.cast(self.arr, ctypes.c_void_p).value
        t.count, t.n_stream = self.count, int(n_stream)
        t.beta1, t.beta2, t.eps = self.key
        t.sched, t.sched_rows = self.sched.data_ptr(), int(self.sched.shape[0])
        return t

    def launch(self, step_dev_ptr: int) -> None:
        b1, b2, eps = self.key
        _lib.check(_lib.load().nic_adam_multi_dev(self.arr, self.count, b1, b2, eps, _lib.ptr(self.sched), int(self.sched.shape[0]),
                                                  ctypes.c_void_p(step_dev_ptr), _lib.stream_ptr(self.device)), "nic_adam_multi_dev")

    def commit(self, n_steps: int) -> None:
        for p in self.params:
            self.opt.state[p]["step"] += n_steps
        self.opt._zeroed = self.zeroed
        self.opt._cache = None


class CosineAnnealing:
    """``torch.optim.lr_scheduler.CosineAnnealingLR`` for the training loop, without its per-step Python machinery (70 us of a 230 us
    host loop): the same chainable recurrence evaluated in the same order in double precision, so the learning rates are bit-identical to
    torch's (tests/test_host_cpu.py holds it to that over whole schedules).  ``step()`` after every optimiser step, like the reference
    (image_compression.py:365, 267)."""

    def __init__(self, optimizer: torch.optim.Optimizer, T_max: int, eta_min: float = 0.0):
        self.optimizer = optimizer
        self.T_max = int(T_max)
        self.eta_min = float(eta_min)
        self.base_lrs = [float(g["lr"]) for g in optimizer.param_groups]
        for g in optimizer.param_groups:
            g.setdefault("initial_lr", g["lr"])
        self.last_epoch = 0

    def get_last_lr(self):
        return [g["lr"] for g in self.optimizer.param_groups]

    def step(self) -> None:
        import math
        self.last_epoch += 1
        t, T = self.last_epoch, self.T_max
        if (t - 1 - T) % (2 * T) == 0:
            for base_lr, g in zip(self.base_lrs, self.optimizer.param_groups):
                g["lr"] = g["lr"] + (base_lr - self.eta_min) * (1 - math.cos(math.pi / T)) / 2
        else:
            f = (1 + math.cos(math.pi * t / T)) / (1 + math.cos(math.pi * (t - 1) / T))
            for g in self.optimizer.param_groups:
                g["lr"] = f * (g["lr"] - self.eta_min) + self.eta_min

    def peek(self, n: int):
        """the learning rates of the next ``n`` optimiser steps (entry 0 = the current ones), per group, without stepping: the very recurrence of
        ``step`` on local copies - bit-identical to stepping"""
        import math
        lrs = [float(g["lr"]) for g in self.optimizer.param_groups]
        t, T = self.last_epoch, self.T_max
        out = []
        for _ in range(n):
            out.append(list(lrs))
            t += 1
            if (t - 1 - T) % (2 * T) == 0:
                lrs = [lr + (base_lr - self.eta_min) * (1 - math.cos(math.pi / T)) / 2 for lr, base_lr in zip(lrs, self.base_lrs)]
            else:
                f = (1 + math.cos(math.pi * t / T)) / (1 + math.cos(math.pi * (t - 1) / T))
                lrs = [f * (lr - self.eta_min) + self.eta_min for lr in lrs]
        return out

    def advance(self, n: int) -> None:
        """``n`` calls of ``step``"""
        for _ in range(n):
            self.step()

    def state_dict(self):
        return {"T_max": self.T_max, "eta_min": self.eta_min, "base_lrs": list(self.base_lrs), "last_epoch": self.last_epoch}

    def load_state_dict(self, sd) -> None:
        self.T_max, self.eta_min, self.base_lrs, self.last_epoch = int(sd["T_max"]), float(sd["eta_min"]), list(sd["base_lrs"]), int(sd["last_epoch"])
