"""Adam as the reference's training loops use it (``torch.optim.Adam(params, lr=0.1)``: voltron/train_utils.py:43,100,
166,238,291), as TWO launches per step -- one torch.cat that gathers the gradients, one HIP kernel (csrc/adam.hip) that
updates every parameter -- with the step count on the device, for the graph-captured loops of train_utils: torch's
capturable Adam is 13 multi-tensor launches per step, and at the reference's sizes an iteration is launch-bound."""
from __future__ import annotations

import struct

import torch

from . import _lib


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam's update (amsgrad=False, weight_decay=0, maximize=False) over every parameter that has a
    gradient at the FIRST step; the set is then fixed (a captured graph replays the same two launches)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._params = None

    def _build(self):
        ps = [p for g in self.param_groups for p in g["params"] if p.grad is not None]
        if not ps:
            return False
        g0 = self.param_groups[0]
        for g in self.param_groups:
            if (g["lr"], g["betas"], g["eps"]) != (g0["lr"], g0["betas"], g0["eps"]):
                raise ValueError("FusedAdam: one set of hyper-parameters for all groups")
        dev = ps[0].device
        rec, end = b"", 0
        for p in ps:
            if p.dtype != torch.float32 or not p.is_contiguous() or p.device != dev:
                raise ValueError("FusedAdam: contiguous fp32 parameters on one device")
            st = self.state[p]
            st["exp_avg"] = torch.zeros_like(p)
            st["exp_avg_sq"] = torch.zeros_like(p)
            end += p.numel()
            rec += struct.pack("<QQQq", p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), end)
        self._slots = torch.frombuffer(bytearray(rec), dtype=torch.uint8).to(dev)
        self._state = torch.zeros(2, dtype=torch.int32, device=dev)
        self._flat = torch.empty(end, dtype=torch.float32, device=dev)
        self._params, self._total = ps, end
        return True

    def extra_state_tensors(self):
        """State outside self.state (train_utils._Snapshot saves / restores it in place): the device step counter."""
        return [] if self._params is None else [self._state]

    @torch.no_grad()
    def step(self, closure=None):
        if self._params is None and not self._build():
            return None
        grads = []
        for p in self._params:
            if p.grad is None:
                raise RuntimeError("FusedAdam: a parameter of the first step has no gradient now")
            grads.append(p.grad.reshape(-1))
        torch.cat(grads, out=self._flat)
        g = self.param_groups[0]
        _lib.check(_lib.lib().volt_adam_step_f32(self._slots.data_ptr(), len(self._params), self._total, self._flat.data_ptr(),
                                                 float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
                                                 self._state.data_ptr(), _lib.stream_ptr()), "volt_adam_step")
        return None
