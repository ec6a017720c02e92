"""Fused Adam over the flat parameter arenas (a8 of SURVEY.md §8) -- torch.optim.Adam semantics
(spec: oracle/colvo_spec.py ADAM_KW), one HIP launch per network instead of one per tensor.
"""
from __future__ import annotations

from typing import Iterable, List

import torch

from . import _lib, ops
from .nn import _ArenaModule


class FusedAdam:
    """opt = FusedAdam([depth_net, pose_net], lr=1e-4); opt.zero_grad(); loss.backward(); opt.step()."""

    def __init__(self, modules: Iterable[_ArenaModule], lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0, zero_grad_in_step: bool = False):
        """zero_grad_in_step (not torch's semantics -- off unless asked for): step() leaves the gradient arenas ZERO (the update
        kernel clears each gradient as it reads it), and the zero_grad() that opens the next step costs no launch while nothing has
        touched the gradients in between.  The loop `zero_grad(); backward(); step()` computes the same thing either way; code
        that reads .grad AFTER step() must leave it off."""
        if weight_decay != 0.0:
            raise NotImplementedError("FusedAdam: weight decay is not on the ColVO path")
        self.modules: List[_ArenaModule] = list(modules)
        for m in self.modules:
            if not isinstance(m, _ArenaModule):
                raise TypeError("FusedAdam takes the HIP-backed networks (coivo_amd.nn.DepthNet / PoseNet)")
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.grad_scale = 1.0
        # A second gradient factor that only exists on the device, for ONE step: ddp.GradBuckets(defer_loss_normalisation=True) posts the
        # 1-element tensor its two-float all-reduce ends in (world / max(3 n_valid of the whole batch, 1)); step() multiplies it into
        # grad_scale inside the update kernel and forgets it.  None: no such factor.
        self.grad_scale_dev = None
        self.zero_grad_in_step = bool(zero_grad_in_step)
        self._multi = _lib.dev_env("COLVO_NO_MULTI_ARENA") is None          # developer A/B switches (COLVO_DEV=1)
        self._fused_pack = _lib.dev_env("COLVO_NO_ADAM_PACK") is None
        # Step numbers live on the host (one launch per network); while a hipGraph is being captured the device counters of
        # the state are used instead (a captured step number would repeat at every replay) and kept in step with the host's.
        self._t = 0
        self.state = []
        for m in self.modules:
            dev = m.flat_param.device
            self.state.append(dict(exp_avg=torch.zeros_like(m.flat_param), exp_avg_sq=torch.zeros_like(m.flat_param),
                                   step=torch.zeros(1, dtype=torch.int32, device=dev)))

    def zero_grad(self, set_to_none: bool = False) -> None:
        # arena semantics: the gradients stay attached; every network's arena is cleared by ONE launch
        if not self._multi:
            for m in self.modules:
                m.zero_grad()
            return
        for m in self.modules:
            m.join_side()
        # (a captured step always holds its clearing launch: a replay cannot know what happened to the arenas since the last one)
        if not (all(m._grads_clean for m in self.modules) and not torch.cuda.is_current_stream_capturing()):
            ops.zero_multi([m.flat_grad for m in self.modules])
        for m in self.modules:
            m.attach_grads()
            m._grads_clean = True           # (cleared just now or known clean; the next backward pass takes it from here --
                                            #  captured: the graph holds the clearing launch in front of every replayed backward)

    @torch.no_grad()
    def step(self) -> None:
        capturing = torch.cuda.is_current_stream_capturing()
        if capturing and not self._device_steps:
            raise RuntimeError("FusedAdam: call use_device_step_counter() before capturing a step into a graph")
        self._t += 1
        if self._fused_pack and self._pack_step():
            return
        if self.grad_scale_dev is not None:
            raise RuntimeError("FusedAdam: a device-side gradient scale needs the one-pass update (colvo_adam_pack_step_scaled: "
                               "networks of one compute dtype, COLVO_NO_ADAM_PACK unset)")
        if self._multi and not self._device_steps and len(self.modules) <= _lib.MAX_ARENAS:
            # one launch for all networks (the step number comes from the host)
            for m in self.modules:
                m.join_side()
                m.attach_grads()
            ops.adam_step_multi([(m.flat_param, m.flat_grad, st["exp_avg"], st["exp_avg_sq"]) for m, st in zip(self.modules, self.state)],
                                self._t, lr=self.lr, beta1=self.betas[0], beta2=self.betas[1], eps=self.eps, grad_scale=self.grad_scale)
            for m in self.modules:
                m.mark_params_changed()
            return
        for m, st in zip(self.modules, self.state):
            m.join_side()
            m.attach_grads()
            if self._device_steps:
                ops.adam_step(m.flat_param, m.flat_grad, st["exp_avg"], st["exp_avg_sq"], st["step"], lr=self.lr,
                              beta1=self.betas[0], beta2=self.betas[1], eps=self.eps, grad_scale=self.grad_scale)
            else:
                ops.adam_step_t(m.flat_param, m.flat_grad, st["exp_avg"], st["exp_avg_sq"], self._t, lr=self.lr,
                                beta1=self.betas[0], beta2=self.betas[1], eps=self.eps, grad_scale=self.grad_scale)
            m.mark_params_changed()

    def writes_operand_copies(self) -> bool:
        """True when step() is the one-pass form (colvo_adam_pack_step) that leaves every network's operand copies current: a
        captured step then needs no repacking node (graph.GraphedTrainStep packs once, eagerly, before the capture)."""
        return self._fused_pack and len({m.compute_dtype for m in self.modules}) == 1

    # ---- update + operand copies in one pass (include/colvo.h colvo_adam_pack_step) -------------------------------------- #
    def _pack_step(self) -> bool:
        """One launch: Adam over every network's arena AND the bf16 / transposed operand copies of the updated 3x3 weights, so
        that the next forward pass finds them current.  False when the networks do not share one compute dtype."""
        dts = {m.compute_dtype for m in self.modules}
        if len(dts) != 1:
            return False
        dt = dts.pop()
        for m in self.modules:
            m.join_side()
            m.attach_grads()
        tab = self._pack_table()
        step_ptr = _lib.ptr(self.state[0]["step"]) if self._device_steps else 0
        gdev, self.grad_scale_dev = self.grad_scale_dev, None          # (one step only)
        if gdev is None:
            _lib.check(_lib.load().colvo_adam_pack_step(ops.dt_code(dt), _lib.ptr(tab[0]), tab[1], tab[2], self.lr, self.betas[0],
                                                        self.betas[1], self.eps, self.grad_scale, step_ptr, self._t, _lib.stream_ptr()),
                       "colvo_adam_pack_step")
        else:
            _lib.check(_lib.load().colvo_adam_pack_step_scaled(ops.dt_code(dt), _lib.ptr(tab[0]), tab[1], tab[2], self.lr, self.betas[0],
                                                               self.betas[1], self.eps, self.grad_scale, _lib.ptr(gdev), step_ptr, self._t,
                                                               _lib.stream_ptr()), "colvo_adam_pack_step_scaled")
        # (with device-side step numbers only the first network's counter is advanced here; _sync_step_state() brings the
        # others in line when the state is read)
        # (while the step is being CAPTURED nothing has run: the arenas are as dirty as they were -- a capture that fails, or one
        # without warm-up steps, must not leave the flag set; graph.GraphedTrainStep sets it after a replay.  ADVICE r4)
        clean = self.zero_grad_in_step and not torch.cuda.is_current_stream_capturing()
        for m in self.modules:
            m.operands_written()
            m._grads_clean = clean
        return True

    def _pack_table(self):
        import numpy as np
        def current_key():      # every address the table holds: a rebuilt arena / operand buffer / moment tensor invalidates it
            return tuple((m.flat_param.data_ptr(), m.flat_grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                          m.compute_dtype, self.zero_grad_in_step) + tuple(None if getattr(m, a, None) is None else getattr(m, a).data_ptr()
                                                   for a in ("_op_bwd", "_op_fwd"))
                         for m, st in zip(self.modules, self.state))
        if self._pack_cache is not None and all(m._pack_table is not None for m in self.modules) and self._pack_cache[0] == current_key():
            return self._pack_cache[1]
        ent = np.dtype([("param", "<u8"), ("grad", "<u8"), ("exp_avg", "<u8"), ("exp_avg_sq", "<u8"), ("fwd", "<u8"), ("bwd", "<u8"),
                        ("w_off", "<i8"), ("fwd_off", "<i8"), ("bwd_off", "<i8"), ("n", "<i8"),
                        ("Cout", "<i4"), ("kk", "<i4"), ("Cin", "<i4"), ("blk", "<i4"), ("kind", "<i4"), ("zero_grad", "<i4")])
        rows, blk = [], 0
        for m, st in zip(self.modules, self.state):
            layers, rest, op_fwd, op_bwd = m.operand_layout()
            base = (m.flat_param.data_ptr(), m.flat_grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                    0 if op_fwd is None else op_fwd.data_ptr(), op_bwd.data_ptr())
            zg = int(self.zero_grad_in_step)
            for w_off, fwd_off, bwd_off, cout, cin in layers:
                rows.append(base + (w_off, fwd_off, bwd_off, 0, cout, 9, cin, blk, 0, zg))
                blk += 9 * ((cout + 31) // 32) * ((cin + 63) // 64)
            for off, n in rest:
                rows.append(base + (off, -1, 0, n, 0, 0, 0, blk, 1, zg))
                blk += (n + _lib.ADAM_PLAIN_PER_WG - 1) // _lib.ADAM_PLAIN_PER_WG
        tab = np.array(rows, dtype=ent)
        dev_tab = torch.from_numpy(tab.view(np.uint8).copy()).to(self.modules[0].flat_param.device)
        self._pack_cache = (current_key(), (dev_tab, len(rows), blk))
        return self._pack_cache[1]

    _pack_cache = None
    _device_steps = False

    def use_device_step_counter(self) -> None:
        """Switch to the device-side step counters (needed when steps are replayed from a hipGraph: graph.py)."""
        if not self._device_steps:
            for st in self.state:
                st["step"].fill_(self._t)
            self._device_steps = True

    def _sync_step_state(self) -> None:
        if self._device_steps:
            self._t = int(self.state[0]["step"].item()) if self.state else self._t
            for st in self.state[1:]:
                st["step"].fill_(self._t)
        else:
            for st in self.state:
                st["step"].fill_(self._t)

    def state_dict(self):
        self._sync_step_state()
        return dict(lr=self.lr, betas=self.betas, eps=self.eps,
                    state=[{k: v.clone() for k, v in st.items()} for st in self.state])

    def load_state_dict(self, sd) -> None:
        self.lr, self.betas, self.eps = sd["lr"], tuple(sd["betas"]), sd["eps"]
        for st, src in zip(self.state, sd["state"]):
            for k in st:
                st[k].copy_(src[k])
        self._t = int(self.state[0]["step"].item()) if self.state else 0
