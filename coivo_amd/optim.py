"""Fused Adam over the flat parameter arenas (a8 of SURVEY.md §8) -- torch.optim.Adam semantics
(spec: oracle/colvo_spec.py ADAM_KW), one HIP launch per network instead of one per tensor.
"""
from __future__ import annotations

from typing import Iterable, List

import torch

from . import ops
from .nn import _ArenaModule


class FusedAdam:
    """opt = FusedAdam([depth_net, pose_net], lr=1e-4); opt.zero_grad(); loss.backward(); opt.step()."""

    def __init__(self, modules: Iterable[_ArenaModule], lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0):
        if weight_decay != 0.0:
            raise NotImplementedError("FusedAdam: weight decay is not on the ColVO path")
        self.modules: List[_ArenaModule] = list(modules)
        for m in self.modules:
            if not isinstance(m, _ArenaModule):
                raise TypeError("FusedAdam takes the HIP-backed networks (coivo_amd.nn.DepthNet / PoseNet)")
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.grad_scale = 1.0
        self.state = []
        for m in self.modules:
            dev = m.flat_param.device
            self.state.append(dict(exp_avg=torch.zeros_like(m.flat_param), exp_avg_sq=torch.zeros_like(m.flat_param),
                                   step=torch.zeros(1, dtype=torch.int32, device=dev)))

    def zero_grad(self, set_to_none: bool = False) -> None:
        for m in self.modules:
            m.zero_grad()

    @torch.no_grad()
    def step(self) -> None:
        for m, st in zip(self.modules, self.state):
            m.join_side()
            m.attach_grads()
            ops.adam_step(m.flat_param, m.flat_grad, st["exp_avg"], st["exp_avg_sq"], st["step"], lr=self.lr,
                          beta1=self.betas[0], beta2=self.betas[1], eps=self.eps, grad_scale=self.grad_scale)
            m.mark_params_changed()

    def state_dict(self):
        return dict(lr=self.lr, betas=self.betas, eps=self.eps,
                    state=[{k: v.clone() for k, v in st.items()} for st in self.state])

    def load_state_dict(self, sd) -> None:
        self.lr, self.betas, self.eps = sd["lr"], tuple(sd["betas"]), sd["eps"]
        for st, src in zip(self.state, sd["state"]):
            for k in st:
                st[k].copy_(src[k])
