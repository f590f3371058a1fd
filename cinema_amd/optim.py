"""AdamW for the hot loop: a torch.optim.Optimizer whose step() is ONE HIP launch over all parameter
groups (cinema_adamw_step).  Same update rule and defaults as torch.optim.AdamW, which the reference
uses (build_atlas.py:453); torch LR schedulers (CosineAnnealingLR, build_atlas.py:455-457) work on it.
Parameters must be contiguous fp32 device tensors."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


class HipAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, clear_grads=False):
        """clear_grads: step() leaves the gradients zeroed (step + the next zero_grad in one pass over them)."""
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._lib = _lib.load()
        self.clear_grads = bool(clear_grads)

    @torch.no_grad()
    def step(self, closure=None):
        entries = []
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous() or not p.is_cuda:
                    raise _lib.CinemaHipError("HipAdamW needs contiguous fp32 parameters and gradients on the GPU")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                g = _lib.AdamWGroup()
                g.p, g.g, g.m, g.v = p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
                g.n = p.numel()
                g.lr, (g.beta1, g.beta2) = float(group["lr"]), group["betas"]
                g.eps, g.weight_decay, g.step = float(group["eps"]), float(group["weight_decay"]), int(st["step"])
                entries.append((g, p.device))
        for i in range(0, len(entries), _lib.ADAMW_MAX_GROUPS):
            chunk = entries[i:i + _lib.ADAMW_MAX_GROUPS]
            arr = (_lib.AdamWGroup * len(chunk))(*[e[0] for e in chunk])
            stream = C.c_void_p(torch.cuda.current_stream(chunk[0][1]).cuda_stream)
            fn = self._lib.cinema_adamw_step_clear if self.clear_grads else self._lib.cinema_adamw_step
            _lib.check(fn(arr, len(chunk), stream), "cinema_adamw_step")
        return None
