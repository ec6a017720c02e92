"""Data-parallel gradient exchange for the DCDP step (a9 of SURVEY.md §8): one process per GPU,
bucketed all-reduce over RCCL/xGMI overlapped with the rest of backward.

The reference has no distributed code; frame pairs are independent units, so the only
exchange is the sum of parameter gradients.  Design for xGMI (point-to-point links, per-link bound):
  * gradients already live in ONE flat fp32 arena per network (coivo_amd.nn), so a bucket is a slice:
    no flatten / unflatten copies, the all-reduce runs in place;
  * the networks report each finished layer (reverse arena order); as soon as the finished suffix of an
    arena crosses a bucket boundary its all-reduce is issued asynchronously (RCCL orders it after the
    kernels already enqueued on the compute stream) while backward keeps running;
  * few, large buckets (default 16 MiB ~ 3 buckets for the 38 MB of gradients): each collective pays the
    ring latency once; the last, non-overlappable one is cut down to the final 1 MiB (the encoder stem), the
    rest of that bucket goes out as soon as its layers are done;
  * optional bf16 transport halves the bytes on the links (BASELINE configs[4]).
`finish()` makes the compute stream wait for all of them; the 1/world_size average is folded into the
fused Adam kernel (grad_scale) instead of a separate pass.
"""
from __future__ import annotations

import os

from typing import List, Optional, Sequence

import torch
import torch.distributed as dist


class _ArenaState:
    def __init__(self, module, bucket_elems: int, tail_elems: int = 0):
        self.module = module
        n = module.flat_grad.numel()
        # bucket boundaries measured from the END of the arena (backward finishes the tail first)
        bounds = list(range(n, 0, -bucket_elems)) + [0]
        # The LAST bucket (front of the arena = the first layers, whose gradients finish last) is the only collective
        # that cannot overlap with backward: keep it small by splitting off the final `tail_elems` when it is larger.
        if tail_elems and bounds[-2] > 2 * tail_elems:
            bounds.insert(len(bounds) - 1, tail_elems)
        self.bounds = bounds            # descending: n, n-b, n-2b, ..., (tail), 0
        self.next = 1                   # index of the next boundary to cross
        self.low = n                    # everything in [low, n) is final
        self.calls = 0
        self.staging = None             # persistent transport-dtype copy of the arena (reduced-precision transport only)


class GradBuckets:
    """Bucketed, overlapped gradient all-reduce over the flat arenas of `modules`.

    modules: objects with `.flat_grad` (1-D tensor) and a settable `.grad_ready_hook(module, begin, end)`
    that they call once per finished layer, in reverse arena order, during backward.
    """

    def __init__(self, modules: Sequence, process_group=None, bucket_bytes: int = 16 << 20,
                 transport_dtype: Optional[torch.dtype] = None, tail_bytes: int = 1 << 20, exact_batch_loss: bool = True):
        """exact_batch_loss (default): the photometric loss is normalised by the valid-pixel count of the WHOLE batch, as the spec
        does (oracle/SPEC.md section 5), not per rank: functional.photometric_loss all-reduces two floats (valid pixels, masked sum)
        right after its forward kernel and every rank scales its raw gradients by world / max(3 n_global, 1) -- data parallel then
        IS the spec's big-batch step (tests/ddp_gpu_worker.py compares with the oracle's plain batch loss).  False: the mean of the
        per-rank masked means (rounds 1-4), which differs when the ranks' valid-pixel counts do; saves one tiny collective between
        forward and backward.  The widened objective (dcdp_full_loss) keeps per-rank normalisers either way."""
        if not dist.is_initialized():
            raise RuntimeError("GradBuckets needs an initialised torch.distributed process group")
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.exact_batch_loss = bool(exact_batch_loss)
        self._queue_claim = None
        if any(getattr(m, "flat_grad", None) is not None and m.flat_grad.is_cuda for m in modules):
            # The communicator brings a stream of its own: main + weight-gradient side stream + communicator is as many
            # active hardware queues as this GPU schedules well.  With the library's second side stream on top the step measured
            # 5.2 ms instead of 1.7 (bench.py --rccl-single): the claim keeps the weight gradients on ONE
            # side stream while this object is attached (streams.py; detach() gives the queue back).
            from . import _lib, nn, streams
            if dist.get_backend(process_group) == "nccl":
                streams.check_environment(self.world)
            # Four hardware queues are served at a time (streams.py): main + ONE shared weight-gradient stream + the library's
            # auxiliary one + RCCL's.  (Round 3 kept a side stream per network and switched the auxiliary one off instead: the
            # one-rank RCCL step 1.55 ms, now 1.39; developer switch COLVO_DDP_OWN_SIDE=1.)
            self._shared_modules = None
            if _lib.dev_env("COLVO_DDP_OWN_SIDE") is None:
                nn.share_side_stream(modules)
                self._shared_modules = list(modules)
            if _lib.dev_env("COLVO_DDP_KEEP_AUX") is None:        # developer A/B switch (COLVO_DEV=1)
                self._queue_claim = streams.claim_external_queue("rccl")
        self.transport_dtype = transport_dtype
        self.states: List[_ArenaState] = []
        self._pending = []
        for m in modules:
            st = _ArenaState(m, max(1, bucket_bytes // 4), max(0, tail_bytes // 4))
            if transport_dtype is not None and transport_dtype != m.flat_grad.dtype:
                # ONE persistent staging arena per network: nothing is allocated inside the hooks (they run on the
                # weight-gradient side stream; a per-call temporary would be consumed by the collective's stream and by the
                # main stream's copy-back, with the caching allocator free to hand it out again in between)
                st.staging = torch.empty(m.flat_grad.numel(), device=m.flat_grad.device, dtype=transport_dtype)
            self.states.append(st)
            m.grad_ready_hook = self._make_hook(st)
        self.attached = True
        self._install_reducer(True)
        # The collectives go to the backend object directly (ProcessGroup.allreduce) instead of through torch.distributed.all_reduce:
        # the Python wrapper's argument checks, rank lookups and logging hooks are ~10 us per call on the host, and an 8-pair step
        # issues seven of them from inside its backward pass, where the host is the bottleneck (one-rank RCCL step at 8 pairs 1.39 ms
        # against 1.30 plain; at 32 / 64 pairs the overhead is 2.2 / 1.6 %).
        # (the group object is looked up per call, not kept: a reference held here outlives destroy_process_group() and the backend's
        # threads are then torn down at interpreter exit -- "terminate called without an active exception", one gloo worker in three)
        self._sum = dist.AllreduceOptions()
        self._sum.reduceOp = dist.ReduceOp.SUM

    @staticmethod
    def _cast(src: torch.Tensor, dst: torch.Tensor, to_transport: bool) -> None:
        """fp32 gradient slice <-> transport-dtype staging slice.  bf16 on the GPU: the library's own kernel (round 4 used
        Tensor.copy_, i.e. at::native kernels inside the data-parallel step); anything else (CPU rehearsals, other dtypes): copy_."""
        bf = dst if to_transport else src
        f32 = src if to_transport else dst
        if (src.is_cuda and bf.dtype == torch.bfloat16 and f32.dtype == torch.float32 and src.data_ptr() % 16 == 0
                and dst.data_ptr() % 16 == 0):
            from . import _lib
            _lib.check(_lib.load().colvo_cast_f32_bf16(_lib.ptr(src), _lib.ptr(dst), src.numel(), int(to_transport), _lib.stream_ptr()),
                       "colvo_cast_f32_bf16")
        else:
            dst.copy_(src)

    def _all_reduce(self, t: torch.Tensor):
        pg = self.group if self.group is not None else dist.distributed_c10d._get_default_group()
        return pg.allreduce([t], self._sum)

    # ---- the spec's batch normalisation under data parallelism ------------------------------------ #
    def _install_reducer(self, on: bool) -> None:
        from . import functional
        if on and self.exact_batch_loss:
            functional.set_batch_reducer(self._reduce_loss_state)
        elif functional._batch_reducer == self._reduce_loss_state:
            functional.set_batch_reducer(None)

    def _reduce_loss_state(self, state: torch.Tensor) -> None:
        """state = {loss, 1/max(3n,1), n_valid, masked sum} of this rank's forward (device): -> the whole batch's."""
        from . import _lib
        # (also with one rank: `bench.py --rccl-single` then prices this collective like the bucket ones, and the state comes back
        # bit for bit -- a one-rank sum is the identity and the rescale repeats the forward's own arithmetic)
        self._all_reduce(state[2:4]).wait()                                           # (the caller's stream waits for it)
        _lib.check(_lib.load().colvo_warp_loss_rescale(_lib.ptr(state), self.world, _lib.stream_ptr()), "colvo_warp_loss_rescale")

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world

    # ---- hook side ----------------------------------------------------------------------------- #
    def _make_hook(self, st: _ArenaState):
        def hook(module, begin: int, end: int) -> bool:
            """Returns True when this call launched a collective (the networks use it to learn where a recorded backward
            has to be split: only there must the hook run between two segments of the command list)."""
            st.calls += 1
            launched = False
            if end == st.module.flat_grad.numel() and st.next > 1:
                raise RuntimeError("GradBuckets: a network ran backward twice in one step after its buckets were "
                                   "already reduced; batch the inputs into one forward (e.g. cat(tgt, ref))")
            if end < st.low:            # not contiguous with the finished suffix: defer to finish()
                return False
            st.low = min(st.low, begin)
            while st.next < len(st.bounds) and st.low <= st.bounds[st.next]:
                self._launch(st, st.bounds[st.next], st.bounds[st.next - 1])
                st.next += 1
                launched = True
            return launched
        return hook

    def trace_buckets(self, on: bool = True) -> None:
        """Measurement aid (bench.py): while on, every bucket launch records a timing event on the stream it is issued from, just
        in front of the collective; bucket_trace(start) then lists where in the step each collective went out -- a first multi-GPU
        record shows the overlap (or its absence) without a profiler."""
        self._trace = [] if on else None

    def bucket_trace(self, start: "torch.cuda.Event"):
        """[{network, bytes, issued_ms after `start`}] of the buckets launched since trace_buckets(True), in launch order; call after
        finish() and a device synchronisation."""
        return [{"network": name, "bytes": nbytes, "issued_ms": round(start.elapsed_time(e0), 4)} for name, nbytes, e0 in (self._trace or [])]

    _trace = None

    def _launch(self, st: _ArenaState, lo: int, hi: int) -> None:
        if hi <= lo:
            return
        if self._trace is not None and not torch.cuda.is_current_stream_capturing():
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
            self._trace.append((type(st.module).__name__, (hi - lo) * st.module.flat_grad.element_size(), e0))
        self._launch_bucket(st, lo, hi)

    def _launch_bucket(self, st: _ArenaState, lo: int, hi: int) -> None:
        sl = st.module.flat_grad[lo:hi]
        if st.staging is not None:
            buf = st.staging[lo:hi]
            self._cast(sl, buf, True)    # conversion on the stream the hook runs on; the collective is ordered after it
            self._pending.append((self._all_reduce(buf), sl, buf))
        else:
            self._pending.append((self._all_reduce(sl), None, None))

    # ---- step side ----------------------------------------------------------------------------- #
    def finish(self) -> None:
        """Issue whatever has not been reduced yet and wait (stream-wise) for every bucket."""
        for st in self.states:
            join = getattr(st.module, "join_side", None)
            if join is not None:
                join()                          # weight gradients whose join the network deferred (nn._ArenaModule)
        for st in self.states:
            while st.next < len(st.bounds):     # layers that reported out of order / never reported
                self._launch(st, st.bounds[st.next], st.bounds[st.next - 1])
                st.next += 1
        for work, sl, buf in self._pending:
            work.wait()
            if buf is not None:
                self._cast(buf, sl, False)
        self._pending.clear()
        for st in self.states:
            st.next, st.low, st.calls = 1, st.module.flat_grad.numel(), 0

    def pause(self) -> None:
        """Take the hooks off the networks (steps then run without any gradient exchange) but keep the communicator's hardware
        queue claimed: the communicator still exists.  resume() puts them back."""
        self._paused_hooks = [st.module.grad_ready_hook for st in self.states]
        for st in self.states:
            st.module.grad_ready_hook = None
        self.attached = False
        self._install_reducer(False)

    def resume(self) -> None:
        for st, h in zip(self.states, getattr(self, "_paused_hooks", [])):
            st.module.grad_ready_hook = h
        self.attached = True
        self._install_reducer(True)

    def detach(self) -> None:
        """Unhook from the networks and give the communicator's hardware queue back to the stream policy."""
        for st in self.states:
            st.module.grad_ready_hook = None
        self.attached = False
        self._install_reducer(False)
        if getattr(self, "_shared_modules", None):
            from . import nn
            nn.unshare_side_stream(self._shared_modules)      # each network back on its own weight-gradient stream
            self._shared_modules = None
        if self._queue_claim is not None:
            self._queue_claim.release()
            self._queue_claim = None
