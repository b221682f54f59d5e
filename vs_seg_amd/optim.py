"""Fused Adam over the model's flat parameter buffer: one HIP launch instead of 178 per-tensor updates.

Semantics = torch.optim.Adam(params, lr, weight_decay) exactly as constructed at ref:params/VSparams.py:388-391 (coupled
L2 decay, bias correction, eps outside the sqrt-of-v̂); `param_groups[i]["lr"]` stays mutable for the halving schedule
(ref:params/VSparams.py:517-523) and `zero_grad()` / `step()` keep their meaning (ref :457, :462).
"""
from __future__ import annotations

import torch

from . import _lib as L


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._owners = []
        for group in self.param_groups:
            owners = {id(getattr(p, "_vsseg_owner", None)): getattr(p, "_vsseg_owner", None) for p in group["params"]}
            if None in owners.values() or len(owners) != 1:
                raise ValueError("vs_seg_amd.optim.Adam updates the flat parameter buffer of one vs_seg_amd model per group; use torch.optim.Adam for other parameters")
            owner = next(iter(owners.values()))
            if len(group["params"]) != len(owner._params):
                raise ValueError("pass model.parameters() (all of them) to vs_seg_amd.optim.Adam")
            self._owners.append(owner)
        self.grad_scale = 1.0  # multiplied into the gradient inside the kernel (data-parallel mean without an extra pass)

    def _flat_state(self, group, flat):
        """m / v / step of one group live in `Optimizer.state` under the group's first parameter (flat tensors spanning the whole
        parameter buffer), so `state_dict()` / `load_state_dict()` carry them like torch.optim.Adam's per-tensor moments."""
        st = self.state[group["params"][0]]
        m = st.get("exp_avg")
        if m is None or m.numel() != flat.numel():
            st["step"], st["exp_avg"], st["exp_avg_sq"] = 0, torch.zeros_like(flat), torch.zeros_like(flat)
        elif m.device != flat.device or m.dtype != flat.dtype:  # the model moved (or a checkpoint was loaded on another device): keep the moments
            st["exp_avg"], st["exp_avg_sq"] = m.to(flat), st["exp_avg_sq"].to(flat)
        if torch.is_tensor(st["step"]):
            st["step"] = int(st["step"])
        return st

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        lib = L.lib()
        stream = torch.cuda.current_stream().cuda_stream
        for group, owner in zip(self.param_groups, self._owners):
            flat, gflat = owner.flat_parameters()
            st = self._flat_state(group, flat)
            if all(p.grad is None for p in group["params"]):
                continue
            st["step"] += 1
            b1, b2 = group["betas"]
            t = st["step"]
            L.check(lib.vsseg_adam(flat.data_ptr(), gflat.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), flat.numel(), float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                   float(group["weight_decay"]), 1.0 - b1**t, 1.0 - b2**t, float(self.grad_scale), stream), "adam")
            torch.autograd.graph.increment_version(flat)  # the kernel wrote through the raw pointer: let version-keyed caches (eval weight packing) see it
        return loss
