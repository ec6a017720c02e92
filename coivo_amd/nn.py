"""HIP-backed DepthNet / PoseNet with the spec's nn.Module signatures and state_dict keys.

Concept: /root/reference/README.md:5,7 (DCDP depth + pose networks, "multimodal fusion" coupling).
The reference ships no code; the layer stack, names and shapes are the frozen spec's
(oracle/colvo_spec.py DepthNet / PoseNet, oracle/SPEC.md §3), so `load_state_dict` works both ways.

MI355X-first host design
  * ParamArena: every parameter of a network is a strided view into ONE flat fp32 buffer (weights in
    [Cout][kh][kw][Cin] memory order = the forward GEMM operand layout), gradients likewise in one flat
    buffer -> one fused Adam launch, and data-parallel all-reduce runs in place on bucket slices.
  * A whole network is ONE autograd node: forward enqueues its kernels back to back on the current
    stream, backward walks the layers in reverse, writing weight gradients straight into the arena
    and reporting each finished layer to an optional hook (gradient buckets for data parallel).
  * Feature maps are NHWC in `compute_dtype` (torch.float32 = exact-f32 MFMA parity mode,
    torch.bfloat16 = throughput mode, fp32 accumulation in both).
  * Each pass (forward / backward of a network at one shape) is RECORDED once as a command list with persistent
    activation buffers and replayed by one native call (program.py, colvo_run_commands): driven layer by layer from
    Python the batch-8 step was bound by the host's ~18 us per launch, not by the GPU.  COLVO_NO_PROGRAM=1 keeps the
    layer-by-layer path (same code, not recorded).
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Tuple

import os

import numpy as np
import torch
import torch.nn as nn

from . import _lib, ops
from .program import Program

# weight-gradient stream: high priority (-1) -- it is the longer of the two chains of the backward pass (measured ~0.4 %
# of the step against normal priority, 0)
_SIDE_PRIORITY = int(_lib.dev_env("COLVO_SIDE_PRIORITY", "-1"))

ENC_CH = (32, 64, 128, 256, 512)
DEC_CH = (16, 32, 64, 128, 256)
POSE_CH = (16, 32, 64, 128, 256, 256, 256)


class ConvParams(nn.Module):
    """weight [Cout, Cin, k, k] (logical, OIHW like nn.Conv2d) + bias [Cout]; memory is arena-owned."""

    def __init__(self, cin: int, cout: int, k: int, cin_pad: Optional[int] = None):
        super().__init__()
        self.cin, self.cout, self.k = cin, cout, k
        self.cin_pad = cin_pad or cin
        self.weight = nn.Parameter(torch.zeros(cout, k, k, cin).permute(0, 3, 1, 2))
        self.bias = nn.Parameter(torch.zeros(cout))
        # filled by the arena
        self.w_master: Optional[torch.Tensor] = None   # [Cout, k*k, cin_pad] fp32 view (arena)
        self.g_master: Optional[torch.Tensor] = None
        self.g_bias: Optional[torch.Tensor] = None
        self.w_fwd: Optional[torch.Tensor] = None      # operand copies in compute dtype
        self.w_bwd: Optional[torch.Tensor] = None
        self.span: Tuple[int, int] = (0, 0)            # [begin, end) of this layer inside the arena


_shared_side = {}


def _side_stream(device) -> "torch.cuda.Stream":
    """The weight-gradient side stream of one network.  Developer A/B COLVO_SHARED_SIDE_STREAM=1: ONE per device, shared by every
    network.  Measured (round 4, 8 pairs): a stream of its own per network is a hardware queue of its own and only four queues are
    served at a time -- with main + two side streams + the library's auxiliary stream the limit is reached, a THIRD weight-gradient
    stream (COLVO_SIDE_STREAMS=3) takes the step from 1.34 to 3.5 ms whatever GPU_MAX_HW_QUEUES says, with the shared stream it
    does not (1.344 ms) -- but it does not win either (two: 1.333), the eager step is level (1.333 against 1.336), and the replayed
    hipGraph lands its side branch on the main chain's queue (2.8-4.0 ms against 1.44; two more streams created in front of the
    capture cure that: graph.py COLVO_GRAPH_SPACER_STREAMS).  Hence: per network, as before."""
    if _lib.dev_env("COLVO_SHARED_SIDE_STREAM") is None:
        return torch.cuda.Stream(device=device, priority=_SIDE_PRIORITY)
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    st = _shared_side.get(key)
    if st is None:
        st = _shared_side[key] = torch.cuda.Stream(device=device, priority=_SIDE_PRIORITY)
    return st


def share_side_stream(modules) -> Optional["torch.cuda.Stream"]:
    """Put the weight gradients of every network in `modules` on ONE side stream (they keep running beside the main stream; the
    networks' backward passes follow one another anyway) and tell the stream policy: a hardware queue is then free for an external
    party -- ddp.GradBuckets calls this for RCCL's stream (streams.py has the measurements)."""
    from . import streams
    nets = [m for m in modules if isinstance(m, _ArenaModule) and m.flat_param.is_cuda]
    if not nets or torch.cuda.is_current_stream_capturing():
        return None
    shared = next((m._side for m in nets if m._side is not None), None)
    if shared is None:
        shared = torch.cuda.Stream(device=nets[0].flat_param.device, priority=_SIDE_PRIORITY)
    for m in nets:
        if m._side is not shared:
            m.join_side()
            if m._side is not None:
                shared.wait_stream(m._side)        # (work a hook may still have in flight there)
            if not hasattr(m, "_side_before_sharing"):
                m._side_before_sharing = m._side   # what unshare_side_stream() gives back (created early: it keeps its hardware queue)
            m._side = shared
    streams.networks_share_side_stream(True)
    return shared


def unshare_side_stream(modules) -> None:
    """Undo share_side_stream(): every network gets the side stream it had before (ddp.GradBuckets.detach() left the
    networks on one stream and the policy's flag set; streams.configure() / reset() cleared the flag while the networks still shared).
    A network that had none yet creates its own at its next backward pass."""
    from . import streams
    if torch.cuda.is_current_stream_capturing():
        return
    for m in modules:
        if isinstance(m, _ArenaModule) and hasattr(m, "_side_before_sharing"):
            m.join_side()
            old = m.__dict__.pop("_side_before_sharing")
            if old is not None and m._side is not None and old is not m._side:
                old.wait_stream(m._side)           # (work still in flight on the shared stream stays ahead of this network's next pass)
            m._side = old
    streams.networks_share_side_stream(False)


class _ArenaModule(nn.Module):
    """Base: owns the flat parameter / gradient arenas of its ConvParams children."""

    def __init__(self, compute_dtype: torch.dtype):
        super().__init__()
        if compute_dtype not in (torch.float32, torch.bfloat16):
            raise TypeError("compute_dtype must be torch.float32 or torch.bfloat16")
        self.compute_dtype = compute_dtype
        self.flat_param: Optional[torch.Tensor] = None
        self.flat_grad: Optional[torch.Tensor] = None
        self._packed_version = None
        self._manual_version = 0
        self._pack_table = None
        self._pack_dtype = None
        self._side = None
        self._plans = {}
        self._insts: Dict[tuple, List["_PassInst"]] = {}     # recorded programs + their activation buffers, per shape
        self.use_programs = _lib.dev_env("COLVO_NO_PROGRAM") is None           # developer switches: COLVO_DEV=1 only
        self._rec: Optional[Program] = None
        self.overlap_wgrad = _lib.dev_env("COLVO_NO_OVERLAP") is None
        # defer_join: the main stream does not wait for this network's weight gradients at the end of ITS backward node but
        # at the end of the whole backward pass (an autograd-engine callback), so the nodes that follow are enqueued in
        # between.  PoseNet's backward sits on the critical path in front of DepthNet's: only its input-gradient chain has
        # to finish before DepthNet's backward may start; its weight gradients then run beside DepthNet's kernels.
        self.defer_join = _lib.dev_env("COLVO_NO_DEFER_JOIN") is None
        self._join_pending = False
        self._cap_side_open = False          # under hipGraph capture: a pass of this network left its side chain open (carry mode)
        self.grad_ready_hook: Optional[Callable[["_ArenaModule", int, int], None]] = None
        # deterministic: weight gradients through per-split slabs + fixed-order second launches instead of float atomics
        # (include/colvo.h colvo_conv_wgrad_det): bitwise repeatable steps at the price of ~one small launch per layer.  Read when a
        # pass is RECORDED: call clear_programs() after changing it.
        self.deterministic = os.environ.get("COLVO_DETERMINISTIC") is not None
        # group_wgrad (round 4; OFF by default): the conv layers' weight gradients leave their kernels as per-split slabs (plain
        # stores, no float atomics) and ONE launch per group of layers adds the slabs to the gradient arena (include/colvo.h
        # colvo_conv_wgrad_slabs / colvo_wgrad_reduce_group).  Built on the in-kernel stamps of profiles/r4_wgrad_phases.md -- the
        # atomics are 6-7.7 us of every weight-gradient launch, the slab stores 2.3 -- and a LOSS in the step: 1.491-1.506 ms against
        # 1.420-1.427 with the atomics at configs[1] (+5 %), 3.80-3.83 against 3.73-3.76 at the configs[3] shape (+2 %), worse with
        # more splits (512-workgroup grids: 1.59 ms) and no better with groups of 3 / 12 / 64 MB; the per-layer deterministic form is
        # +1.6 %.  The slabs are 180 MB a step written and read back through HBM beside kernels that are bandwidth-bound at the top
        # of the network, and a group's launch waits for BOTH weight-gradient streams; the atomics are slow per launch but ride in
        # L2 under the other streams' kernels.  A group closes when it holds wgrad_group_bytes of gradients (or 16 layers); its
        # layers are reported to the gradient-ready hook then, in backward order.  Kept as an option (weight gradients are bitwise
        # repeatable in it; `deterministic` stays the per-layer form, which is the cheaper of the two).  Read when a pass is RECORDED.
        self.group_wgrad = _lib.dev_env("COLVO_WGRAD_GROUPED") is not None
        self.wgrad_group_bytes = int(_lib.dev_env("COLVO_WGRAD_GROUP_MB", "6")) << 20
        self._group: list = []
        self._group_bytes = 0

    # ---- arena ------------------------------------------------------------------------------- #
    def _layers(self) -> List[ConvParams]:
        # the module tree is fixed after construction: walk it once (per step this list and the parameter list below were
        # ~0.3 ms of the host's 1.1 ms enqueue time, tools/host_profile.py)
        ll = self.__dict__.get("_layer_list")
        if ll is None:
            ll = [m for m in self.modules() if isinstance(m, ConvParams)]
            self.__dict__["_layer_list"] = ll
            self.__dict__["_param_list"] = list(self.parameters())
        return ll

    def _build_arena(self, device) -> None:
        layers = self._layers()
        total = 0
        for L in layers:
            total += L.cout * L.k * L.k * L.cin_pad + L.cout
            total = (total + 63) // 64 * 64         # keep every layer 256-byte aligned
        flat = torch.zeros(total, device=device, dtype=torch.float32)
        grad = torch.zeros(total, device=device, dtype=torch.float32)
        off = 0
        for L in layers:
            nw = L.cout * L.k * L.k * L.cin_pad
            begin = off
            wv = flat[off:off + nw].view(L.cout, L.k, L.k, L.cin_pad)
            gv = grad[off:off + nw].view(L.cout, L.k, L.k, L.cin_pad)
            with torch.no_grad():
                wv[..., :L.cin].copy_(L.weight.detach().permute(0, 2, 3, 1).to(device))
            L.weight.data = wv[..., :L.cin].permute(0, 3, 1, 2)
            L.weight.grad = None
            L.w_master = flat[off:off + nw].view(L.cout, L.k * L.k, L.cin_pad)
            L.g_master = grad[off:off + nw].view(L.cout, L.k * L.k, L.cin_pad)
            L._gw_view = gv[..., :L.cin].permute(0, 3, 1, 2)
            off += nw
            with torch.no_grad():
                flat[off:off + L.cout].copy_(L.bias.detach().to(device))
            L.bias.data = flat[off:off + L.cout]
            L.bias.grad = None
            L.g_bias = grad[off:off + L.cout]
            off += L.cout
            off = (off + 63) // 64 * 64
            L.span = (begin, off)
            L.w_fwd = L.w_bwd = None
        self.flat_param, self.flat_grad = flat, grad
        self._packed_version = None
        self._pack_table = None
        self._insts = {}            # recorded programs hold pointers into the old arena

    def _apply(self, fn, *a, **kw):   # .to() / .cuda(): rebuild the arena on the new device
        super()._apply(fn, *a, **kw)
        p = next(self.parameters())
        self._build_arena(p.device)
        return self

    # True while the gradient arena is known to be all zero (set by optim.FusedAdam(zero_grad_in_step=True).step(), cleared by every
    # backward pass): the optimizer's zero_grad() then skips its launch
    _arena_zero_for_pass = False        # _grads_clean as the running backward pass found it (handed to its weight-gradient commands)
    _clean_flag = False
    _clean_version = -1
    _clean_in_capture = False

    # "The gradient arena is all zero."  Setting the flag records the arena's version counter: every torch-side write to a .grad view
    # (AccumulateGrad for a parameter that is also used outside the fused network, a manual p.grad.add_) bumps it, the library's own
    # kernels -- raw pointers -- do not; the flag reads True only while the counter is where the clearing left it.  Set while a stream
    # is being captured (the graph holds the clearing launch, nothing has executed) it holds inside that capture only: a capture that
    # fails, or ends, leaves it False.  (the flag alone was unguarded, and single-split layers STORE into an arena they
    # believe to be zero; an eager zero_grad() skips its launch on it.)
    @property
    def _grads_clean(self) -> bool:
        if not self._clean_flag or self.flat_grad is None:
            return False
        if self._clean_in_capture:
            return torch.cuda.is_current_stream_capturing()
        return self.flat_grad._version == self._clean_version

    @_grads_clean.setter
    def _grads_clean(self, clean: bool) -> None:
        self._clean_flag = bool(clean)
        if clean and self.flat_grad is not None:
            self._clean_version = self.flat_grad._version
            self._clean_in_capture = self.flat_grad.is_cuda and torch.cuda.is_current_stream_capturing()

    def _take_arena_zero(self) -> None:
        """Opens a backward pass: 'the arena is still zero' goes to its weight-gradient commands, and stops being true."""
        self._arena_zero_for_pass = self._grads_clean
        self._clean_flag = False

    def attach_grads(self) -> None:
        """Point every p.grad at its arena view (zeroing the arena if grads were set to None)."""
        layers = self._layers()
        for p in self.__dict__["_param_list"]:          # (the Parameter objects themselves: no nn.Module attribute lookups)
            if p.grad is None:
                break
        else:
            return
        ops.zero_(self.flat_grad)
        for L in layers:
            L.weight.grad = L._gw_view
            L.bias.grad = L.g_bias

    def zero_grad(self, set_to_none: bool = False) -> None:   # arena semantics: grads stay attached
        self.join_side()
        if self.flat_grad is not None:
            ops.zero_(self.flat_grad)          # one hipMemsetAsync, no torch fill kernel
            self.attach_grads()
            self._grads_clean = True           # until a backward pass -- or a torch-side write to .grad (version counter) -- touches it

    def mark_params_changed(self) -> None:
        self._manual_version += 1

    def _weights_version(self):
        # in-place updates of the parameters (any optimizer, load_state_dict) bump their version counters
        self._layers()
        return (sum(p._version for p in self.__dict__["_param_list"]), self._manual_version, self.compute_dtype)

    def _prepare_weights(self) -> None:
        """Refresh the operand-layout copies of the weights when the master changed (one launch per network)."""
        ver = self._weights_version()
        if ver == self._packed_version:
            return
        self._ensure_operand_buffers()
        ops.pack_weights_multi(self.flat_param, self._pack_table, self._pack_n, self._pack_blocks, self.compute_dtype, self._op_fwd,
                               self._op_bwd)
        self._packed_version = ver

    def operand_layout(self):
        """For an optimizer that writes the operand copies itself (optim.FusedAdam -> colvo_adam_pack_step):
        -> (layers, rest, op_fwd, op_bwd): layers = [(w_off, fwd_off or -1, bwd_off, Cout, Cin_pad)] of the 3x3 layers whose
        weights have operand copies, rest = [(offset, n)] every other range of the arena, the two operand buffers."""
        self._ensure_operand_buffers()
        layers, rest, at = [], [], 0
        for w_off, fwd_off, bwd_off, cout, cin in self._pack_layers:
            if w_off > at:
                rest.append((at, w_off - at))
            layers.append((w_off, fwd_off, bwd_off, cout, cin))
            at = w_off + cout * 9 * cin
        if self.flat_param.numel() > at:
            rest.append((at, self.flat_param.numel() - at))
        return layers, rest, self._op_fwd, self._op_bwd

    def operands_written(self) -> None:
        """The caller has just rewritten the master weights AND both operand copies (see operand_layout)."""
        self._manual_version += 1
        self._packed_version = self._weights_version()

    def _ensure_operand_buffers(self) -> None:
        dt = self.compute_dtype
        dev = self.flat_param.device
        if self._pack_table is None or self._pack_dtype != dt:
            layers = [L for L in self._layers() if L.k == 3 and L.cout > 1]   # heads read the fp32 master directly
            tab = np.zeros(len(layers), dtype=np.dtype([("w_off", "<i8"), ("fwd_off", "<i8"), ("bwd_off", "<i8"),
                                                        ("Cout", "<i4"), ("kk", "<i4"), ("Cin", "<i4"), ("blk", "<i4")]))
            total = sum(L.cout * 9 * L.cin_pad for L in layers)
            self._insts = {}        # recorded programs hold pointers into the old operand copies
            self._op_bwd = torch.empty(total, device=dev, dtype=dt)
            self._op_fwd = None if dt == torch.float32 else torch.empty(total, device=dev, dtype=dt)
            off = blk = 0
            self._pack_layers = []
            for i, L in enumerate(layers):
                n = L.cout * 9 * L.cin_pad
                tab[i] = (L.span[0], -1 if self._op_fwd is None else off, off, L.cout, 9, L.cin_pad, blk)
                self._pack_layers.append((L.span[0], -1 if self._op_fwd is None else off, off, L.cout, L.cin_pad))
                L.w_bwd = self._op_bwd[off:off + n].view(L.cin_pad, 9, L.cout)
                L.w_fwd = L.w_master if self._op_fwd is None else self._op_fwd[off:off + n].view(L.cout, 9, L.cin_pad)
                off += n
                blk += 9 * ((L.cout + 31) // 32) * ((L.cin_pad + 63) // 64)   # include/colvo.h: colvo_pack_weights_multi
            self._pack_table = torch.from_numpy(tab.view(np.uint8).copy()).to(dev)
            self._pack_n, self._pack_blocks, self._pack_dtype = len(layers), blk, dt
            self._packed_version = None

    # ---- recorded passes ------------------------------------------------------------------------------- #
    def clear_programs(self) -> None:
        """Drop the recorded programs and their persistent activation buffers."""
        self._insts = {}

    def _acquire(self, key) -> "_PassInst":
        pool = self._insts.setdefault(key, [])
        for inst in pool:
            if not inst.busy:
                inst.busy = True
                return inst
        inst = _PassInst()
        inst.busy = True
        pool.append(inst)
        return inst

    def _run_pass(self, inst: "_PassInst", which: str, externals: Dict[str, torch.Tensor], body):
        """Run body() -- a sequence of ops.* calls -- as the pass `which` of `inst`: recorded on first use, replayed
        afterwards with the externals' pointers patched in.  Returns what body() returned when it was recorded."""
        if not self.use_programs:
            if torch.cuda.is_current_stream_capturing():
                # the layer-by-layer developer path forks the weight gradients onto a real side stream with events: captured, that
                # is the multi-stream graph that aborted the runtime under GPU_MAX_HW_QUEUES=2 in round 2
                raise RuntimeError("hipGraph capture needs the recorded passes (COLVO_NO_PROGRAM / use_programs=False cannot be captured)")
            return body()
        entry = inst.passes.get(which)
        if entry is None:
            with Program() as pr:
                self._rec = pr
                try:
                    for name, t in externals.items():
                        pr.external(name, t)
                    out = body()
                finally:
                    self._rec = None
            entry = inst.passes[which] = (pr, out)
        else:
            pr = entry[0]
            for name, t in externals.items():
                pr.patch(name, t)
        pr, out = entry
        if pr.flag_slots:
            # "the gradient arena is still zero" for this pass's weight-gradient commands: true for the first backward pass behind an
            # optimizer step that cleared the gradients (FusedAdam(zero_grad_in_step=True)) or an explicit zero_grad()
            pr.set_flags(1 if self._arena_zero_for_pass else 0)
        if pr.uses_side and self._side is None:
            self._side = _side_stream(self.flat_param.device)
        if self.grad_ready_hook is None or not pr.marks:
            pr.run(self._side)
        else:
            # Data parallel: the hook of a finished layer may launch a collective, which must be enqueued right after that
            # layer's kernels.  The first replay splits the program after EVERY layer and notes which hook calls launched
            # something (a hook that returns None is assumed to); later replays split only there and run the hooks of the
            # layers in between together, still in order.  A launch at an unexpected place re-arms the learning pass.
            capturing = torch.cuda.is_current_stream_capturing()

            def call(L, on_side):
                # (under hipGraph capture there is no side stream: the weight-gradient chain the library keeps is joined here --
                # in carry mode a segment does not end joined by itself -- and the collective is captured behind it)
                if on_side and not capturing:
                    with torch.cuda.stream(self._side):
                        return self.grad_ready_hook(self, L.span[0], L.span[1])
                if capturing:
                    _lib.check(_lib.load().colvo_capture_join(_lib.stream_ptr()), "colvo_capture_join")
                return self.grad_ready_hook(self, L.span[0], L.span[1])

            done = 0
            flush = getattr(pr, "flush", None)
            if flush is None or len(flush) != len(pr.marks):
                learned = []
                for idx, (L, on_side) in pr.marks:
                    pr.run(self._side, done, idx)
                    done = idx
                    r = call(L, on_side)
                    learned.append(r is None or bool(r))
                pr.flush = learned
            else:
                pending, ok = [], True
                for i, (idx, mark) in enumerate(pr.marks):
                    pending.append(mark)
                    if flush[i]:
                        pr.run(self._side, done, idx)
                        done = idx
                        for j, (L, on_side) in enumerate(pending):
                            r = call(L, on_side)
                            ok = ok and ((r is None or bool(r)) == (j == len(pending) - 1))
                        pending = []
                if pending:
                    pr.run(self._side, done, pr.marks[-1][0])
                    done = pr.marks[-1][0]
                    for L, on_side in pending:
                        ok = ok and not call(L, on_side)
                if not ok:
                    pr.flush = None
            pr.run(self._side, done, None)
        if getattr(pr, "deferred_join", False):
            if torch.cuda.is_current_stream_capturing():
                self._cap_side_open = True
            self._queue_join()
        return out

    # ---- backward scheduling: weight gradients on a side stream, concurrent with the input gradients ------ #
    def _bwd_begin(self) -> None:
        self._group, self._group_bytes = [], 0
        self._main = torch.cuda.current_stream()
        if self.overlap_wgrad and self._side is None:
            self._side = _side_stream(self.flat_param.device)
        self._side_used = False

    def _run_wgrad(self, L, fn, *tensors) -> None:
        """fn() enqueues weight-gradient kernel(s); with overlap it runs on the side stream, ordered after everything already
        enqueued on the main stream (its inputs), and so does the gradient-ready hook (the data-parallel all-reduce of finished
        buckets) of the layers that are COMPLETE once fn() has run: L = one ConvParams, a list of them, or None."""
        layers = [] if L is None else ([L] if isinstance(L, ConvParams) else list(L))
        rec = self._rec
        if not self.overlap_wgrad:
            fn()
            for l_ in layers:
                self._layer_done(l_)
            return
        if rec is not None:
            rec.fork()
            rec.stream = 1
            fn()
            for l_ in layers:
                self._layer_done(l_)
            rec.stream = 0
            self._side_used = True
            return
        ev = torch.cuda.Event()
        ev.record(self._main)
        self._side.wait_event(ev)
        with torch.cuda.stream(self._side):
            fn()
            for l_ in layers:
                self._layer_done(l_)
        for t in tensors:
            if t is not None:
                t.record_stream(self._side)
        self._side_used = True

    def _conv_wgrad(self, L: ConvParams, desc, x0, x1, dy) -> None:
        """Weight / bias gradient of conv layer L into the gradient arena: atomics (default), the per-layer deterministic form, or grouped slabs (see group_wgrad)."""
        if not self.group_wgrad:
            scr = ops.conv_wgrad_scratch(desc, dy.device) if self.deterministic else None       # (per-layer deterministic form)
            self._run_wgrad(L, lambda: ops.conv_wgrad(desc, x0, x1, dy, L.g_master, L.g_bias, scr), x0, x1, dy)
            return
        scr = ops.conv_wgrad_scratch(desc, dy.device)          # (recorded passes keep it: persistent, like the activations)
        nsplit = ops.conv_wgrad_splits(desc)
        self._run_wgrad(None, lambda: ops.conv_wgrad_slabs(desc, x0, x1, dy, scr), x0, x1, dy, scr)
        self._group.append((L, (scr, L.g_master, L.g_bias, nsplit, desc.Cout, desc.C0 + desc.C1)))
        self._group_bytes += L.g_master.numel() * 4
        if self._group_bytes >= self.wgrad_group_bytes or len(self._group) == _lib.WGRAD_GROUP_MAX:
            self._wgrad_flush()

    def _wgrad_flush(self) -> None:
        """Close the open group: one launch adds its layers' slabs to the arena, then the layers count as done (in order)."""
        if not self._group:
            return
        group, self._group, self._group_bytes = self._group, [], 0
        sets = [g_[1] for g_ in group]

        def fn():
            if self._rec is not None:
                self._rec.side_sync()        # the group's kernels were dealt onto both side streams: wait for all of them
            ops.wgrad_reduce_group(sets)
        self._run_wgrad([g_[0] for g_ in group], fn, *[t for st_ in sets for t in st_[:3]])

    def _bwd_end(self) -> None:
        self._wgrad_flush()
        if self.overlap_wgrad and self._side_used:
            if self.defer_join:
                if self._rec is not None:
                    self._rec.deferred_join = True
                self._queue_join()
            elif self._rec is not None:
                self._rec.join()
            elif not torch.cuda.is_current_stream_capturing():
                self._main.wait_stream(self._side)

    def _queue_join(self) -> None:
        """Join the side stream when the running backward pass ends (falls back to joining now outside the engine)."""
        if self._join_pending or torch.cuda.is_current_stream_capturing():      # (a captured pass ends joined: program.hip)
            return
        self._join_pending = True
        main = torch.cuda.current_stream()

        def _final_join():
            if self._join_pending and self._side is not None:
                main.wait_stream(self._side)
            self._join_pending = False
        try:
            torch.autograd.Variable._execution_engine.queue_callback(_final_join)
        except RuntimeError:              # not inside a backward pass (a backward body driven by hand)
            _final_join()

    def join_side(self) -> None:
        """Make the current stream wait for the weight gradients still running on this network's side stream.  Under hipGraph
        capture the side chain lives in the library (csrc/program.hip, carry mode): the next captured node is made to depend on it."""
        if torch.cuda.is_current_stream_capturing():
            _lib.check(_lib.load().colvo_capture_join(_lib.stream_ptr()), "colvo_capture_join")
            self._cap_side_open = False
        elif self._join_pending and self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)
        self._join_pending = False
        self._stored_pass_open = False

    _stored_pass_open = False            # a backward pass of this network stored (not added) its weight gradients and is not joined yet

    def _order_deterministic_pass(self) -> None:
        """Deterministic mode, a SECOND backward pass of this network while the first one's weight gradients may still be running on
        the side streams (deferred join: one network applied twice inside one backward): the one-split write-out of
        colvo_conv_wgrad_det adds to dw / db with a plain read-modify-write -- sole writer within a launch, not across two launches on
        different streams -- so the earlier pass's side work is joined first.  The default (atomic) form needs no order."""
        # (the same holds for an earlier pass that STORED into the clean arena -- colvo_conv_wgrad_clean -- while this one adds
        #  with atomics: an add that lands before the store is lost.  Such a pass is joined too.)
        if self.deterministic or self._stored_pass_open:
            if torch.cuda.is_current_stream_capturing():
                if self._cap_side_open:          # (carry mode leaves this network's side chain open; policy 3 has two of them)
                    self.join_side()
            elif self._join_pending:
                self.join_side()
        self._stored_pass_open = self._arena_zero_for_pass

    def _layer_done(self, L: ConvParams) -> None:
        if self._rec is not None:
            self._rec.mark((L, self._rec.stream == 1))
        elif self.grad_ready_hook is not None:
            self.grad_ready_hook(self, L.span[0], L.span[1])

    def _trigger(self) -> torch.Tensor:
        # a leaf that requires grad so that autograd schedules the network's backward node
        t = getattr(self, "_trig", None)
        if t is None or t.device != self.flat_param.device:
            t = torch.zeros(1, device=self.flat_param.device, requires_grad=True)
            self._trig = t
        return t


def _dealias(tensors):
    """Externals of a recorded pass are matched by address: two per-call tensors that alias (pose_net(x, x); autograd
    handing one tensor as two gradients) would be indistinguishable, so later duplicates get a private copy."""
    seen, out = set(), []
    for t in tensors:
        if t is not None and t.data_ptr() in seen:
            t = t.clone()
        if t is not None:
            seen.add(t.data_ptr())
        out.append(t)
    return out


class _PassInst:
    """The recorded passes of one network at one shape, with the activation buffers they refer to.  `busy` from a
    forward that may still get a backward until that backward ran (or its autograd context died): a second forward
    in between gets its own instance instead of overwriting the saved activations."""

    def __init__(self):
        self.busy = False
        self.passes: Dict[str, tuple] = {}
        self.saved = None
        self.gen = 0             # bumped by every forward run on this instance: a hand-over of one of its buffers names the run


class _Lease:
    def __init__(self, inst: Optional[_PassInst]):
        self.inst = inst

    def release(self) -> None:
        if self.inst is not None:
            self.inst.busy = False
            self.inst = None

    def __del__(self):
        self.release()


def _conv(x0, L: ConvParams, desc, x1=None):
    y = torch.empty(desc.B, desc.Ho, desc.Wo, desc.Cout, device=x0.device, dtype=x0.dtype)
    ops.conv_fwd(desc, x0, x1, L.w_fwd, L.bias.data, y)
    return y


class DepthNet(_ArenaModule):
    """forward(img [B,3,H,W] fp32) -> depth [B,1,H,W] fp32 in (0.1, 10).  H, W multiples of 32."""

    def __init__(self, compute_dtype: torch.dtype = torch.float32, device="cuda"):
        super().__init__(compute_dtype)
        cin = 3
        for i, c in enumerate(ENC_CH, start=1):
            setattr(self, f"enc{i}a", ConvParams(cin, c, 3, cin_pad=8 if cin == 3 else cin))
            setattr(self, f"enc{i}b", ConvParams(c, c, 3))
            cin = c
        for i in range(5, 0, -1):
            d = DEC_CH[i - 1]
            setattr(self, f"up{i}", ConvParams(cin, d, 3))
            skip = ENC_CH[i - 2] if i >= 2 else 0
            setattr(self, f"iconv{i}", ConvParams(d + skip, d, 3))
            cin = d
        self.head = ConvParams(cin, 1, 3)
        self._build_arena(torch.device(device))

    def forward(self, img: torch.Tensor) -> torch.Tensor:
        """depth [B,1,H,W].  For an even batch -- the DCDP pair batch `depth_net(torch.cat([tgt, ref]))` of the spec's train step --
        the result is a _PairDepth: an ordinary depth tensor for every use, except that the spec's own next line, `d[:B]` /
        `d[B:]` with B = half the batch, returns the two halves as OUTPUTS of the network's autograd node (as forward_pair_split
        does) instead of slices of one output: no slice-backward nodes (zero-fill + copy + add each), PoseNet's input can come
        from the pass, and photometric_loss(.., d[:B], ..) finds the loss's own gradient path (functional.photometric_loss)."""
        # (the pair form exists for the backward pass: under no_grad / inference_mode -- where tensors carry no version counter for
        #  its tags, and the extra PoseNet-input buffer would only be kept alive -- the plain node runs.  ADVICE r5)
        if (img.dim() == 4 and img.shape[0] % 2 == 0 and img.shape[0] >= 2 and self.use_programs
                and torch.is_grad_enabled() and not torch.is_inference_mode_enabled()
                and _lib.dev_env("COLVO_NO_PAIR_FORWARD") is None):
            from .functional import GradHandover
            hand = GradHandover()
            d_t, d_r, d_l, full = _DepthNetPairFn.apply(self, img, self._trigger(), hand, False, True)
            d_l._colvo_handover = hand
            self._tag_pose_in((d_t, d_r))
            d_t._colvo_loss_twin = (d_l, d_l._version)      # photometric_loss swaps it in (once): the loss's own gradient path
            return _PairDepth.wrap(full, d_t, d_r)
        return _DepthNetFn.apply(self, img, self._trigger())

    def forward_pair(self, frames: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """frames [2B,3,H,W] = target frames then reference frames -> (depth_t, depth_r), each [B,1,H,W].
        Same as `d = self(frames); d[:B], d[B:]`, but the two halves are outputs of ONE autograd node: the backward
        gets their gradients directly instead of through two slice-backward nodes (zero-fill + copy + add each, ~8
        small launches on the critical path between the PoseNet and the DepthNet backward)."""
        if frames.dim() != 4 or frames.shape[0] % 2:
            raise ValueError("forward_pair: expected [2B,3,H,W]")
        return self._tag_pose_in(_DepthNetPairFn.apply(self, frames, self._trigger(), None)[:2])

    def forward_pair_split(self, frames: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """forward_pair with a third output: the target-frame depth AGAIN, as the tensor to hand to photometric_loss (and
        to nothing else).  It aliases depth_t, but being an output of its own the backward node receives the loss
        gradient separately from PoseNet's: no autograd add, no concatenation, and -- through the GradHandover it carries --
        no normalisation pass: the fused loss returns its unnormalised depth gradient plus two device scalars and the
        head backward applies them while it reads the gradient (functional.GradHandover)."""
        if frames.dim() != 4 or frames.shape[0] % 2:
            raise ValueError("forward_pair_split: expected [2B,3,H,W]")
        from .functional import GradHandover
        hand = GradHandover()
        d_t, d_r, d_l = _DepthNetPairFn.apply(self, frames, self._trigger(), hand)
        d_l._colvo_handover = hand
        self._tag_pose_in((d_t, d_r))
        return d_t, d_r, d_l

    def forward_pair_full(self, frames: torch.Tensor):
        """forward_pair with BOTH depths a second time -> (depth_t, depth_r, depth_t', depth_r'): the primed pair goes to an
        objective that consumes both frames' depth (functional.dcdp_full_loss) and to nothing else, the plain pair to
        PoseNet.  The backward node then receives the four gradients separately and the head's backward kernel adds them
        while it reads them: no autograd accumulation, zero-fill or concatenation kernels."""
        if frames.dim() != 4 or frames.shape[0] % 2:
            raise ValueError("forward_pair_full: expected [2B,3,H,W]")
        return self._tag_pose_in(_DepthNetPairFn.apply(self, frames, self._trigger(), None, True))

    def _tag_pose_in(self, outs):
        """The target-frame depth carries the PoseNet input this pass filled (see _forward_impl) to PoseNet.forward."""
        outs[0]._colvo_pose_in = getattr(self, "_pose_in", None)
        self._pose_in = None
        return outs

    # ---- whole-network forward / backward ---------------------------------------------------- #
    def _plan(self, B, H, W):
        dt = self.compute_dtype
        key = (B, H, W, dt)
        if key in self._plans:
            return self._plans[key]
        P = self._plans[key] = {}
        h, w, cin = H, W, 8
        for i, c in enumerate(ENC_CH, start=1):
            P[f"enc{i}a"] = ops.conv_desc(dt, B, h, w, cin, c, stride=2)
            h, w = h // 2, w // 2
            P[f"enc{i}b"] = ops.conv_desc(dt, B, h, w, c, c)
            cin = c
        for i in range(5, 0, -1):
            d = DEC_CH[i - 1]
            h, w = h * 2, w * 2
            P[f"up{i}"] = ops.conv_desc(dt, B, h, w, cin, d, up0=True)
            skip = ENC_CH[i - 2] if i >= 2 else 0
            P[f"iconv{i}"] = ops.conv_desc(dt, B, h, w, d, d, C1=skip)
            cin = d
        return P

    def _forward_impl(self, img: torch.Tensor, pair: bool = False):
        """pair: img is the DCDP pair batch [target frames | reference frames]; where the top of the network runs fused
        (colvo_conv_head_fused) the pass then also fills PoseNet's 8-channel input -- rgb from the stem's packing kernel, the two
        depth channels from the head -- and leaves it in self._pose_in for PoseNet.forward (which checks that it is handed these
        very frames and depths, and packs for itself otherwise)."""
        if img.dim() != 4 or img.shape[1] != 3:
            raise ValueError(f"DepthNet: expected [B,3,H,W], got {tuple(img.shape)}")
        B, _, H, W = img.shape
        if H % 32 or W % 32:
            raise ValueError("DepthNet: H and W must be multiples of 32")
        if not img.is_cuda or img.dtype != torch.float32:
            raise RuntimeError("DepthNet: expects a float32 CUDA tensor (no CPU fallback)")
        self._prepare_weights()
        P = self._plan(B, H, W)
        img = img.contiguous()
        depth = torch.empty(B, 1, H, W, device=img.device, dtype=torch.float32)
        fuse_top = ops.conv_head_fused_ok(P["iconv1"]) and _lib.dev_env("COLVO_NO_FWD16") is None
        pose_fill = (pair and fuse_top and B % 2 == 0 and not torch.is_inference_mode_enabled()      # (its tag reads version counters)
                     and _lib.dev_env("COLVO_NO_POSE_FILL") is None)
        inst = self._acquire((B, H, W, self.compute_dtype, pose_fill))

        def body():
            A: Dict[str, torch.Tensor] = {}
            if pose_fill:
                x = torch.empty(B, H, W, 8, device=img.device, dtype=self.compute_dtype)
                # (the two depth channels of every pixel are written by the head below, the six rgb channels here)
                A["pose_in"] = torch.empty(B // 2, H, W, 8, device=img.device, dtype=self.compute_dtype)
                ops.pack_stem_pose(img, x, A["pose_in"])
            else:
                x = ops.pack_nchw([img], 8, self.compute_dtype)
            A["in"] = x
            for i in range(1, 6):
                x = _conv(x, getattr(self, f"enc{i}a"), P[f"enc{i}a"]); A[f"enc{i}a"] = x
                x = _conv(x, getattr(self, f"enc{i}b"), P[f"enc{i}b"]); A[f"enc{i}b"] = x
            for i in range(5, 0, -1):
                x = _conv(x, getattr(self, f"up{i}"), P[f"up{i}"]); A[f"up{i}"] = x
                if i == 1 and fuse_top:
                    # the narrow full-resolution layer and the depth head in one pass: its 42 MB output is written, not read back
                    L = self.iconv1
                    y = torch.empty(P["iconv1"].B, P["iconv1"].Ho, P["iconv1"].Wo, 16, device=x.device, dtype=x.dtype)
                    ops.conv_head_fused(P["iconv1"], x, L.w_fwd, L.bias.data, self.head.w_master, self.head.bias.data, y, depth,
                                        A.get("pose_in"))
                    A["iconv1"] = x = y
                    return A
                x = _conv(x, getattr(self, f"iconv{i}"), P[f"iconv{i}"], A[f"enc{i - 1}b"] if i >= 2 else None)
                A[f"iconv{i}"] = x
            ops.depth_head_fwd(x, self.head.w_master, self.head.bias.data, depth)
            return A

        A = self._run_pass(inst, "fwd", {"img": img, "depth": depth}, body)
        inst.gen += 1
        # (the tag holds `img` itself: while the tag lives the frames' memory cannot be handed to another tensor)
        self._pose_in = (A["pose_in"], img, depth.data_ptr(), depth._version, inst, inst.gen, img._version) if pose_fill else None
        return depth, (A, P, inst)

    def _backward_impl(self, saved, depth: torch.Tensor, d_depth: Optional[torch.Tensor], parts=None) -> None:
        """d_depth [B,1,H,W], or parts = (g_first, g_second, g_raw, scale_a, scale_b[, g_raw_second]): the gradient of the
        first / second half of the images and a (scaled) addend for each half (ops.depth_head_bwd_parts), each may be None."""
        A, P, inst = saved
        self._take_arena_zero()
        self.attach_grads()
        self._order_deterministic_pass()
        B, _, H, W = depth.shape
        dev = depth.device
        if parts is None:
            d_depth = d_depth.contiguous()
            ext = {"depth": depth, "d_depth": d_depth}
            which = "bwd"
        else:
            parts = tuple(_dealias([None if t is None else t.contiguous() for t in parts]))
            names = ("g_first", "g_second", "g_raw", "scale_a", "scale_b", "g_raw_second")
            ext = {"depth": depth}
            ext.update({n: t for n, t in zip(names, parts) if t is not None})
            which = "bwd:" + ",".join(n for n, t in zip(names, parts) if t is not None)

        def body():
            self._bwd_begin()

            def wgrad(name, x0, x1, dy):
                self._conv_wgrad(getattr(self, name), P[name], x0, x1, dy)

            def dgrad(name, src, dy, mask_like, dx=None, accumulate=False):
                L = getattr(self, name)
                if dx is None:
                    dx = torch.empty_like(mask_like)
                ops.conv_dgrad(P[name], src, dy, L.w_bwd, mask_like, dx, accumulate)
                return dx

            x1 = A["iconv1"]
            scratch = torch.empty(B * H * W, device=dev, dtype=torch.float32)
            # The narrow full-resolution layer (iconv1) takes its input gradient and weight gradient in ONE pass over its input and
            # output (include/colvo.h colvo_conv_bwd_fused); in the training step's form of this call (gradient in parts) the head's
            # input gradient is made inside that pass as well (HEAD form) and never touches memory.
            fuse1 = (not self.deterministic and not self.group_wgrad and ops.conv_bwd_fused_ok(P["iconv1"]) and
                     _lib.dev_env("COLVO_NO_BWD16") is None)
            fuse_head = fuse1 and parts is not None and _lib.dev_env("COLVO_NO_BWD16_HEAD") is None
            g = None if fuse_head else torch.empty_like(x1)
            # head: d(pre) + input gradient on the main stream, its weight gradient beside it like every other layer's
            if parts is None:
                ops.depth_head_bwd(x1, self.head.w_master, depth, d_depth, scratch, g, None, None)
            else:
                ops.depth_head_bwd_parts(x1, self.head.w_master, depth, *parts[:5], scratch, g,
                                         parts[5] if len(parts) > 5 else None)
            # (the head's OWN weight gradient can ride along too -- colvo_conv_bwd_fused's head_partials -- and measured no gain: it
            # leaves the side streams, where it overlapped, for the main chain: 1.464 against 1.455 ms at configs[1], 3.729 against
            # 3.732 at the configs[3] shape; developer switch COLVO_BWD16_HEADW=1)
            fuse_headw = fuse_head and _lib.dev_env("COLVO_BWD16_HEADW") is not None
            if not fuse_headw:
                if self.compute_dtype == torch.bfloat16 and _lib.dev_env("COLVO_NO_HEAD_WGRAD_MFMA") is None:
                    # the head's weight gradient by MFMA (partial rows + table reduction: reproducible in every mode)
                    self._run_wgrad(self.head, lambda: ops.depth_head_wgrad_mfma(x1, scratch, self.head.g_master, self.head.g_bias),
                                    x1, scratch)
                else:
                    self._run_wgrad(self.head, lambda: ops.depth_head_wgrad(x1, scratch, self.head.g_master, self.head.g_bias,
                                                                            self.deterministic), x1, scratch)
            d_skip: Dict[int, torch.Tensor] = {}
            for i in range(1, 6):                       # decoder, output side first
                u = A[f"up{i}"]
                skip = A[f"enc{i - 1}b"] if i >= 2 else None
                Pi = P[f"iconv{i}"]
                if i == 1 and fuse1:
                    # on the main stream: d_u is the next layer's dy
                    L = getattr(self, f"iconv{i}")
                    d_u = torch.empty_like(u)
                    if fuse_headw:
                        rows = ops.conv_bwd_fused_head_rows(Pi)
                        hp = torch.empty(rows * 145, device=dev, dtype=torch.float32)
                        ops.conv_bwd_fused(Pi, x1, L.w_bwd, u, True, d_u, L.g_master, L.g_bias, scratch, self.head.w_master, hp)
                        self._run_wgrad(self.head, lambda: ops.depth_head_wgrad_reduce(hp, rows, self.head.g_master, self.head.g_bias), hp)
                    elif fuse_head:
                        ops.conv_bwd_fused(Pi, x1, L.w_bwd, u, True, d_u, L.g_master, L.g_bias, scratch, self.head.w_master)
                    else:
                        ops.conv_bwd_fused(Pi, g, L.w_bwd, u, True, d_u, L.g_master, L.g_bias)
                    self._layer_done(L)
                    below = A[f"iconv{i + 1}"] if i < 5 else A["enc5b"]
                    wgrad(f"up{i}", below, None, d_u)
                    g = dgrad(f"up{i}", 0, d_u, below)
                    continue
                wgrad(f"iconv{i}", u, skip, g)
                if skip is not None:                    # both sources' input gradients in one launch
                    L = getattr(self, f"iconv{i}")
                    d_u, d_skip[i - 1] = torch.empty_like(u), torch.empty_like(skip)
                    ops.conv_dgrad_both(P[f"iconv{i}"], g, L.w_bwd, u, skip, d_u, d_skip[i - 1])
                else:
                    d_u = dgrad(f"iconv{i}", 0, g, u)
                below = A[f"iconv{i + 1}"] if i < 5 else A["enc5b"]
                wgrad(f"up{i}", below, None, d_u)
                g = dgrad(f"up{i}", 0, d_u, below)
            for i in range(5, 0, -1):                   # encoder
                a = A[f"enc{i}a"]
                wgrad(f"enc{i}b", a, None, g)
                g_a = dgrad(f"enc{i}b", 0, g, a)
                src = A["in"] if i == 1 else A[f"enc{i - 1}b"]
                wgrad(f"enc{i}a", src, None, g_a)
                if i > 1:
                    g = dgrad(f"enc{i}a", 0, g_a, src, dx=d_skip[i - 1], accumulate=True)
            self._bwd_end()

        self._run_pass(inst, which, ext, body)


class _PairDepth(torch.Tensor):
    """What DepthNet.forward returns for an even batch: the depth of the whole batch, an ordinary tensor in every respect (ops on
    it return plain tensors), whose two half-batch slices `d[:B]` / `d[B:]` are the pre-split outputs of the network's node."""
    __torch_function__ = torch._C._disabled_torch_function_impl

    @staticmethod
    def wrap(full: torch.Tensor, d_t: torch.Tensor, d_r: torch.Tensor) -> "_PairDepth":
        t = full.as_subclass(_PairDepth)
        t._colvo_halves = (full.shape[0] // 2, d_t, d_r, full._version)
        return t

    def __getitem__(self, idx):
        h = self.__dict__.get("_colvo_halves")
        if h is not None and isinstance(idx, slice) and idx.step in (None, 1) and self._version == h[3]:
            B, d_t, d_r, _ = h
            start, stop, _ = idx.indices(2 * B)
            if (start, stop) == (0, B):
                return d_t
            if (start, stop) == (B, 2 * B):
                return d_r
        return super().__getitem__(idx)


class _DepthNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net: DepthNet, img, trigger):
        depth, saved = net._forward_impl(img)
        ctx.net, ctx.saved = net, saved
        ctx.lease = _Lease(saved[2])          # frees the pass instance when this context dies without a backward
        ctx.save_for_backward(depth)
        return depth

    @staticmethod
    def backward(ctx, d_depth):
        (depth,) = ctx.saved_tensors
        ctx.net._backward_impl(ctx.saved, depth, d_depth)
        ctx.saved = None
        ctx.lease.release()
        return None, None, None


class _DepthNetPairFn(torch.autograd.Function):
    """depth of [target frames | reference frames] as THREE outputs of one node: depth_t, depth_r and depth_t once more
    for the loss (forward_pair_split).  The backward receives the three gradients separately and hands them to the head's
    backward kernel as they are (no zero-fill, cat or add kernels in between)."""

    @staticmethod
    def forward(ctx, net: DepthNet, frames, trigger, handover, both=False, whole=False):
        depth, saved = net._forward_impl(frames, pair=True)
        ctx.net, ctx.saved, ctx.handover, ctx.whole = net, saved, handover, whole
        ctx.set_materialize_grads(False)           # an output nobody used arrives as None, not as a tensor of zeros to add
        ctx.lease = _Lease(saved[2])
        ctx.save_for_backward(depth)
        B = frames.shape[0] // 2
        if both:                                   # forward_pair_full: BOTH depths once more, for an objective that takes both
            return depth[:B], depth[B:], depth[:B], depth[B:]
        if whole:                                  # DepthNet.forward: the whole batch as a fourth output (any other use of `d`)
            return depth[:B], depth[B:], depth[:B], depth.view_as(depth)
        return depth[:B], depth[B:], depth[:B]

    @staticmethod
    def backward(ctx, g_t, g_r, g_l, g_4=None):
        (depth,) = ctx.saved_tensors
        sa = sb = None
        if ctx.handover is not None:
            sa, sb = ctx.handover.take((g_l,))     # unnormalised loss gradient + its two device scale factors
        if ctx.whole and g_4 is not None:
            # the whole-batch output was used as well (d.mean(), a slice other than the halves ...): the general backward on the sum
            B = depth.shape[0] // 2
            d_depth = g_4.clone()
            for half, g in ((d_depth[:B], g_t), (d_depth[B:], g_r)):
                if g is not None:
                    half.add_(g)
            if g_l is not None:
                d_depth[:B].add_(g_l * (sa * sb) if sa is not None else g_l)
            ctx.net._backward_impl(ctx.saved, depth, d_depth)
        else:
            ctx.net._backward_impl(ctx.saved, depth, None, parts=(g_t, g_r, g_l, sa, sb, None if ctx.whole else g_4))
        ctx.saved = None
        ctx.lease.release()
        return None, None, None, None, None, None


class PoseNet(_ArenaModule):
    """forward(tgt, ref, tgt_depth=None, ref_depth=None) -> (pose [B,6], lcc_a [B,1], lcc_b [B,1])."""

    def __init__(self, compute_dtype: torch.dtype = torch.float32, device="cuda"):
        super().__init__(compute_dtype)
        cin = 8
        for i, c in enumerate(POSE_CH, start=1):
            setattr(self, f"conv{i}", ConvParams(cin, c, 3))
            cin = c
        self.pred = ConvParams(cin, 8, 1)
        # 7 tiny layers: driven from Python the fork/join bookkeeping costs more than it hides; recorded it is free.
        # With a deferred join (FusedAdam) only the input-gradient chain stays in front of DepthNet's backward.
        self.overlap_wgrad = self.overlap_wgrad and self.use_programs and _lib.dev_env("COLVO_NO_POSE_OVERLAP") is None
        self._build_arena(torch.device(device))

    def forward(self, tgt, ref, tgt_depth: Optional[torch.Tensor] = None, ref_depth: Optional[torch.Tensor] = None):
        from .functional import GradHandover
        hand = GradHandover()
        pose, a, b = _PoseNetFn.apply(self, tgt, ref, tgt_depth, ref_depth, self._trigger(), hand)
        # handed DIRECTLY to photometric_loss, the three outputs get their gradients unnormalised plus two device scalars
        # which the head's backward kernel applies (functional.GradHandover); any other use takes the ordinary path
        pose._colvo_handover = a._colvo_handover = b._colvo_handover = hand
        return pose, a, b

    def _forward_impl(self, tgt, ref, d_t, d_r):
        B, _, H, W = tgt.shape
        for t in (tgt, ref):
            if tuple(t.shape) != (B, 3, H, W) or not t.is_cuda or t.dtype != torch.float32:
                raise RuntimeError("PoseNet: expects float32 CUDA images of equal shape [B,3,H,W] (no CPU fallback)")
        if (d_t is None) != (d_r is None):
            raise ValueError("PoseNet: pass both depth maps or neither")
        for t in (d_t, d_r):
            if t is not None and tuple(t.shape) != (B, 1, H, W):
                raise ValueError("PoseNet: depth maps must be [B,1,H,W]")
        self._prepare_weights()
        dt = self.compute_dtype
        has_depth = d_t is not None
        # DepthNet's pair pass may have assembled this very input already (DepthNet._forward_impl): taken only when tgt, ref, d_t
        # and d_r are, untouched, the frames that pass read and the depths it wrote
        filled = getattr(d_t, "_colvo_pose_in", None) if has_depth else None
        if filled is not None:
            # ... and the buffer still holds THAT pass: it belongs to DepthNet's pass instance, which a later forward re-uses as
            # soon as this one's autograd context is gone (at once under no_grad) -- `gen` names the run that filled it -- and
            # the frames have not been written in place since (d1 = forward_pair(f1); d2 = forward_pair(f2);
            # pose_net(f1[:B], f1[B:], *d1) passed the address / depth-version checks with f2's rgb and d2's depth in the buffer)
            buf, frames, depth_ptr, version, src_inst, gen, frames_version = filled
            frames_ptr = frames.data_ptr()
            ok = (tuple(buf.shape) == (B, H, W, 8) and buf.dtype == dt and tgt.is_contiguous() and ref.is_contiguous()
                  and d_t.is_contiguous() and d_r.is_contiguous() and tgt.data_ptr() == frames_ptr
                  and ref.data_ptr() == frames_ptr + 12 * B * H * W and d_t.data_ptr() == depth_ptr
                  and d_r.data_ptr() == depth_ptr + 4 * B * H * W and d_t._version == version and d_r._version == version
                  and src_inst.gen == gen and frames._version == frames_version and tgt._version == frames_version
                  and ref._version == frames_version)
            filled = buf if ok else None
        if filled is not None:
            has_depth = "filled"
            srcs = []
        else:
            srcs = _dealias([tgt.contiguous(), ref.contiguous()] + ([d_t.contiguous(), d_r.contiguous()] if has_depth else []))
        key = (B, H, W, dt)
        P = self._plans.get(key)
        if P is None:
            P = self._plans[key] = {}
            h, w, cin = H, W, 8
            for i, c in enumerate(POSE_CH, start=1):
                P[i] = ops.conv_desc(dt, B, h, w, cin, c, stride=2)
                h, w, cin = P[i].Ho, P[i].Wo, c
        out = torch.empty(8 * B, device=tgt.device, dtype=torch.float32)     # planar: [pose Bx6 | a B | b B]
        inst = self._acquire((B, H, W, dt, has_depth))

        def body():
            x = filled if filled is not None else ops.pack_nchw(srcs, 8, dt)
            A = {"in": x}
            for i in range(1, 8):
                x = _conv(x, getattr(self, f"conv{i}"), P[i])
                A[i] = x
            ops.pose_head_fwd(x, self.pred.w_master, self.pred.bias.data, out)
            return A

        ext = {f"src{k}": t for k, t in enumerate(srcs)}
        ext["out"] = out
        if filled is not None:
            ext["in"] = filled             # belongs to DepthNet's pass instance: may be another buffer at the next step
        A = self._run_pass(inst, "fwd", ext, body)
        return out, (A, P, (B, H, W), has_depth, inst, filled)

    def _backward_impl(self, saved, d_pose, d_a, d_b, scale_a=None, scale_b=None):
        A, P, (B, H, W), has_depth, inst, filled = saved
        self._take_arena_zero()
        self.attach_grads()
        self._order_deterministic_pass()
        dev = self.flat_param.device
        grads = {"d_pose": d_pose, "d_a": d_a, "d_b": d_b, "scale_a": scale_a, "scale_b": scale_b}
        grads = {k: v.contiguous() for k, v in grads.items() if v is not None}
        grads = dict(zip(grads.keys(), _dealias(list(grads.values()))))
        # both depth gradients from one launch: [2][B,1,H,W], each half a contiguous tensor of its own
        d_tr = torch.empty(2, B, 1, H, W, device=dev, dtype=torch.float32) if has_depth else None
        d_t, d_r = (d_tr[0], d_tr[1]) if has_depth else (None, None)

        def body():
            self._bwd_begin()
            x = A[7]
            g = torch.empty_like(x)
            ops.pose_head_bwd(x, self.pred.w_master, grads.get("d_pose"), grads.get("d_a"), grads.get("d_b"),
                              g, self.pred.g_master, self.pred.g_bias, grads.get("scale_a"), grads.get("scale_b"), self.deterministic)
            self._layer_done(self.pred)
            for i in range(7, 0, -1):
                L = getattr(self, f"conv{i}")
                src = (filled if filled is not None else A["in"]) if i == 1 else A[i - 1]
                self._conv_wgrad(L, P[i], src, None, g)
                if i > 1:
                    dx = torch.empty_like(src)
                    ops.conv_dgrad(P[i], 0, g, L.w_bwd, src, dx, False)
                    g = dx
                elif has_depth and _lib.dev_env("COLVO_NO_DGRAD_PLANES") is None:
                    # only the two depth channels of the 8-channel input gradient are wanted, as fp32 planes: one small kernel
                    # instead of the full input gradient + an unpack pass (include/colvo.h colvo_conv_dgrad_planes)
                    ops.conv_dgrad_planes(P[1], g, L.w_master, 6, 2, d_tr)
                elif has_depth:                      # (developer A/B switch: the round-3 form)
                    dx = torch.empty_like(src)
                    ops.conv_dgrad(P[i], 0, g, L.w_bwd, None, dx, False)
                    ops.unpack_nhwc_grad(dx, 6, 2, d_tr, False, by_channel=True)
            self._bwd_end()

        ext = dict(grads)
        if has_depth:
            ext["d_tr"] = d_tr
        if filled is not None:
            ext["in"] = filled
        which = "bwd:" + ",".join(sorted(grads))       # a missing (None) gradient changes the recorded commands
        self._run_pass(inst, which, ext, body)
        return d_t, d_r


class _PoseNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net: PoseNet, tgt, ref, d_t, d_r, trigger, handover):
        out, saved = net._forward_impl(tgt, ref, d_t, d_r)
        ctx.net, ctx.saved, ctx.handover = net, saved, handover
        ctx.lease = _Lease(saved[4])
        B = tgt.shape[0]
        # three contiguous views of the planar head output: no slicing kernels forward or backward
        return out[:6 * B].view(B, 6), out[6 * B:7 * B].view(B, 1), out[7 * B:].view(B, 1)

    @staticmethod
    def backward(ctx, d_pose, d_a, d_b):
        sa, sb = ctx.handover.take((d_pose, d_a, d_b)) if ctx.handover is not None else (None, None)
        d_t, d_r = ctx.net._backward_impl(ctx.saved, d_pose, d_a, d_b, sa, sb)
        ctx.saved = None
        ctx.lease.release()
        return None, None, None, d_t, d_r, None, None


def dcdp_forward(depth_net: DepthNet, pose_net: PoseNet, tgt, ref, K, *, ssim_weight: float = 0.85, full_loss: bool = False,
                 frames: Optional[torch.Tensor] = None):
    """One coupled DCDP forward (spec: dcdp_forward): depth of both frames -> pose + LCC -> loss
    (full_loss: the widened objective dcdp_full_loss -- multi-scale photometric + geometric consistency + smoothness).
    frames (not in the spec): the pair batch already stacked as [target frames | reference frames] = [2B,3,H,W], as the loader
    delivers it; tgt / ref are then ignored (may be None) and the concatenation kernel is saved."""
    from .functional import dcdp_full_loss, photometric_loss
    if frames is not None:
        if frames.dim() != 4 or frames.shape[0] % 2:
            raise ValueError("dcdp_forward: frames must be [2B,3,H,W]")
        tgt, ref = frames[:frames.shape[0] // 2], frames[frames.shape[0] // 2:]
    else:
        frames = torch.cat([tgt, ref], dim=0)
        if not (tgt.requires_grad or ref.requires_grad):
            tgt, ref = frames[:tgt.shape[0]], frames[tgt.shape[0]:]     # the same values; PoseNet then recognises the pair batch
    if full_loss:
        # the objective is one native call that returns finished gradients (no hand-over of scale factors); the depth of both
        # frames reaches it through outputs of their own, so no gradient is accumulated by autograd
        d_t, d_r, d_lt, d_lr = depth_net.forward_pair_full(frames)
        pose, a, b = pose_net(tgt, ref, d_t, d_r)
        loss = dcdp_full_loss(tgt, ref, d_lt, d_lr, pose, K, a, b, ssim_weight=ssim_weight)
        return loss, d_t, d_r, pose, a, b
    d_t, d_r, d_l = depth_net.forward_pair_split(frames)
    pose, a, b = pose_net(tgt, ref, d_t, d_r)
    loss = photometric_loss(tgt, ref, d_l, pose, K, a, b, ssim_weight=ssim_weight)     # d_l aliases d_t (its own grad path)
    return loss, d_t, d_r, pose, a, b
