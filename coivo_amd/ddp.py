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


class _NativeRccl:
    """The process group's RCCL communicator, called through RCCL's own C ABI (round 5).

    ProcessGroupNCCL.allreduce costs ~35 us of host time a call -- work object, event pair, stream bookkeeping, watchdog list --
    and an 8-pair step issues eight collectives from inside its backward pass, where the host is what the GPU waits for: the
    one-rank RCCL step ran +7...30 % over the plain one depending on the box's host.  torch keeps owning the
    communicator (creation, rank layout, teardown); `_comm_ptr()` hands its ncclComm_t over once, and a collective is then ONE
    ctypes call of ncclAllReduce on a stream of ours, ordered by two event operations -- in place, capturable (RCCL supports stream
    capture), invisible to torch's watchdog (no eager work objects: the hazard GraphedTrainStep.capture() guards against is gone).

    Rules this object lives by:
      * ONE stream carries every collective of the communicator, in issue order -- exactly what ProcessGroupNCCL's internal stream
        does -- so NCCL's "operations of one communicator never run concurrently, same order on every rank" holds by construction;
      * torch must not run collectives of its own on this communicator while the object is in use: they would go to torch's internal
        stream (concurrency with ours), and that stream would become a FIFTH active hardware queue -- the step then runs 2.7 x slower
        (streams.py; measured with this very class before it stopped asking torch for a warm-up collective: 3.5 ms instead of 1.3).
        bench.py therefore synchronises its ranks over a gloo side group; barrier() below is the device-side alternative;
      * what is given up: torch's watchdog never sees these collectives, so a hung all-reduce (a dead peer) is neither timed out nor
        aborted by it -- the job hangs until its launcher's own timeout.  The ProcessGroup.allreduce path keeps that coverage."""

    _F32, _BF16, _SUM = 7, 9, 0          # ncclFloat32, ncclBfloat16, ncclSum (nccl.h / rccl.h)

    def __init__(self, group):
        import ctypes as C
        dev = torch.device("cuda", torch.cuda.current_device())
        pg = group if group is not None else dist.distributed_c10d._get_default_group()
        backend = pg._get_backend(dev)
        if not hasattr(backend, "_comm_ptr"):
            raise RuntimeError("this torch build does not expose ProcessGroupNCCL._comm_ptr()")
        try:
            comm = int(backend._comm_ptr())
        except Exception:           # noqa: BLE001 -- no communicator yet (init_process_group without device_id): create it, no collective
            comm = 0
        if not comm:
            backend.eager_connect_single_device(dev)
            comm = int(backend._comm_ptr())
        if not comm:
            raise RuntimeError("ProcessGroupNCCL._comm_ptr() returned NULL")
        self.comm = comm
        # The RCCL torch itself has mapped -- not "a" librccl.so: a second copy of the library (another version behind the bare
        # soname) would be handed torch's ncclComm_t, which is undefined behaviour the self-check below cannot be relied on to catch
        # (ADVICE r5).  /proc/self/maps names the file; RTLD_NOLOAD opens it only if it is mapped already.  Not found: no native path.
        self.lib_path = self._loaded_rccl()
        if self.lib_path is None:
            raise RuntimeError("no librccl mapped into this process (is torch's nccl backend RCCL?)")
        lib = C.CDLL(self.lib_path, mode=getattr(os, "RTLD_NOLOAD", 4) | getattr(os, "RTLD_NOW", 2))
        self._fn = lib.ncclAllReduce
        self._fn.restype = C.c_int
        self._fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        self._err = lib.ncclGetErrorString
        self._err.restype = C.c_char_p
        self._err.argtypes = [C.c_int]
        # What RCCL itself says about the communicator torch handed over, read through the SAME handle the collectives will use: it
        # must be this group's (a first multi-GPU record then shows that the native path saw N ranks).
        n, r = C.c_int(-1), C.c_int(-1)
        for fn, out in ((lib.ncclCommCount, n), (lib.ncclCommUserRank, r)):
            fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.POINTER(C.c_int)]
            rc = fn(self.comm, C.byref(out))
            if rc != 0:
                raise RuntimeError(f"{fn.__name__} failed: {self._err(rc).decode('utf-8', 'replace')} ({rc})")
        self.comm_count, self.comm_rank = int(n.value), int(r.value)
        if (self.comm_count, self.comm_rank) != (dist.get_world_size(group), dist.get_rank(group)):
            raise RuntimeError(f"the communicator says rank {self.comm_rank} of {self.comm_count}, the process group rank "
                               f"{dist.get_rank(group)} of {dist.get_world_size(group)}")
        self.world = dist.get_world_size(group)
        self.stream = torch.cuda.Stream(device=dev)           # the collectives' own queue (the one streams.py budgets for RCCL)
        self._ev = torch.cuda.Event()                         # "everything the collective reads has been enqueued" (re-recorded per call)
        self._ev_back = torch.cuda.Event()
        self.dirty = False                                    # collectives issued on `stream` since the last join()
        # self-check with known values, before anything depends on it (also the communicator's first collective)
        self._probe = torch.full((8,), 1.5, device=dev)
        self.all_reduce(self._probe)
        self.join()
        torch.cuda.synchronize()
        if not torch.equal(self._probe.cpu(), torch.full((8,), 1.5 * self.world)):
            raise RuntimeError(f"native ncclAllReduce self-check failed: {self._probe.tolist()} (world {self.world})")
        if self.comm_rank == 0:
            import logging
            logging.getLogger("coivo_amd.ddp").info("native RCCL path: comm %#x, %d ranks (ncclCommCount), this is rank %d, %s", self.comm,
                                                    self.comm_count, self.comm_rank, self.lib_path)

    @staticmethod
    def _loaded_rccl() -> Optional[str]:
        try:
            with open("/proc/self/maps") as f:
                for line in f:
                    path = line.rsplit(None, 1)[-1] if "/" in line else ""
                    if "librccl" in os.path.basename(path) and os.path.exists(path):
                        return path
        except OSError:
            pass
        return None

    def describe(self) -> dict:
        """What bench.py puts into its JSON line (N > 1): which library, which communicator, what RCCL says about it."""
        return {"librccl": self.lib_path, "comm": hex(self.comm), "ncclCommCount": self.comm_count, "ncclCommUserRank": self.comm_rank}

    def _call(self, t: torch.Tensor) -> None:
        code = self._F32 if t.dtype == torch.float32 else self._BF16 if t.dtype == torch.bfloat16 else None
        if code is None or not t.is_contiguous():
            raise TypeError(f"native all-reduce of a {t.dtype} / non-contiguous tensor")
        rc = self._fn(t.data_ptr(), t.data_ptr(), t.numel(), code, self._SUM, self.comm, self.stream.cuda_stream)
        if rc != 0:
            raise RuntimeError(f"ncclAllReduce failed: {self._err(rc).decode('utf-8', 'replace')} ({rc})")

    def all_reduce(self, t: torch.Tensor) -> None:
        """In-place sum over the ranks on the collectives' stream, ordered after everything enqueued on the CURRENT stream so far;
        join() makes a stream wait for it."""
        self._ev.record()                                     # on the current stream
        self.stream.wait_event(self._ev)
        self._call(t)
        self.dirty = True

    def all_reduce_now(self, t: torch.Tensor) -> None:
        """In-place sum the CURRENT stream waits for at once (the two-float loss state between forward and backward)."""
        was = self.dirty
        self.all_reduce(t)
        self._ev_back.record(self.stream)
        torch.cuda.current_stream().wait_event(self._ev_back)
        self.dirty = was            # (what was pending for join() before still is: join() may run on another stream)

    def join(self) -> None:
        """The current stream waits for every collective issued so far."""
        if self.dirty:
            self._ev_back.record(self.stream)
            torch.cuda.current_stream().wait_event(self._ev_back)
            self.dirty = False

    def barrier(self) -> None:
        """Every rank has reached this point and this rank's device is idle (a one-element all-reduce, then a host wait)."""
        self._probe.fill_(1.0)
        self.all_reduce(self._probe)
        self.join()
        torch.cuda.synchronize()


class GradBuckets:
    """Bucketed, overlapped gradient all-reduce over the flat arenas of `modules`.

    modules: objects with `.flat_grad` (1-D tensor) and a settable `.grad_ready_hook(module, begin, end)`
    that they call once per finished layer, in reverse arena order, during backward.
    """

    def __init__(self, modules: Sequence, process_group=None, bucket_bytes: int = 16 << 20,
                 transport_dtype: Optional[torch.dtype] = None, tail_bytes: int = 1 << 20, exact_batch_loss: bool = True,
                 native_collectives: bool = False, defer_loss_normalisation: bool = False, optimizer=None):
        """exact_batch_loss (default): the photometric loss is normalised by the valid-pixel count of the WHOLE batch, as the spec
        does (oracle/SPEC.md section 5), not per rank: functional.photometric_loss all-reduces two floats (valid pixels, masked sum)
        right after its forward kernel and every rank scales its raw gradients by world / max(3 n_global, 1) -- data parallel then
        IS the spec's big-batch step (tests/ddp_gpu_worker.py compares with the oracle's plain batch loss).  False: the mean of the
        per-rank masked means (rounds 1-4), which differs when the ranks' valid-pixel counts do; saves one tiny collective between
        forward and backward.  The widened objective (dcdp_full_loss) keeps per-rank normalisers either way.
        defer_loss_normalisation=True (needs optimizer=the step's FusedAdam; round 6): the two-float exchange is only STARTED behind the
        loss kernel and waited for in finish().  Between forward and backward no rank waits for another: the backward pass runs on the
        UNNORMALISED loss gradients (every parameter gradient is linear in the normaliser), the buckets sum those, and finish() hands
        the optimizer world / max(3 n_global, 1) as a device-side factor of its gradient scale (FusedAdam.grad_scale_dev).  The loss
        tensor reads the whole batch's value after finish().  An opt-in because it binds the caller: the photometric loss -- ONE call
        per step, taken through the hand-over path (nn.dcdp_forward, forward_pair_split, the spec's own call sequence) -- must be
        the only source of parameter gradients in the step (anything else would be scaled along), and finish() / optimizer.step()
        must follow every backward pass.  Calls that cannot defer (a loss without the hand-over mailboxes) exchange at once, as before.
        photometric_loss's reducer is ONE process-wide hook (functional.set_batch_reducer), installed by the last GradBuckets built or
        resumed: a process that keeps several GradBuckets attached at once (tests/graph_rccl_worker.py does) must re-install the right
        one before each step (_install_reducer(True)) -- with this option the reducer carries state.
        native_collectives=True: RCCL called through its own C ABI on the group's communicator (_NativeRccl: half the host cost per
        collective) instead of through ProcessGroup.allreduce -- for an nccl group with GPU arenas, unless COLVO_DDP_TORCH_COLLECTIVES=1
        is exported.  An opt-in because it binds the CALLER: no torch collective may run on this group from before the attach to the
        detach (it would add torch's internal stream as a fifth hardware queue: steps 2.7 x slower; see _NativeRccl) -- barriers and
        logging reductions go over a gloo side group or GradBuckets.barrier().  bench.py opts in when its gloo group exists."""
        if not dist.is_initialized():
            raise RuntimeError("GradBuckets needs an initialised torch.distributed process group")
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.exact_batch_loss = bool(exact_batch_loss)
        self._opt = optimizer
        self._defer = bool(defer_loss_normalisation and exact_batch_loss and optimizer is not None
                           and os.environ.get("COLVO_DDP_BLOCKING_LOSS", "0") in ("", "0"))
        if defer_loss_normalisation and optimizer is None:
            raise ValueError("GradBuckets(defer_loss_normalisation=True) needs optimizer=<the step's FusedAdam>")
        if self._defer and not (hasattr(optimizer, "writes_operand_copies") and optimizer.writes_operand_copies()):
            raise ValueError("GradBuckets(defer_loss_normalisation=True): the optimizer's one-pass update (colvo_adam_pack_step_scaled) "
                             "is what takes the device-side scale; it is not available for this set of networks")
        self._deferred = None                   # (state, work or None) of the exchange started in this step's forward pass
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
        self._eager_works = []
        self._install_reducer(True)
        # The collectives go to the backend object directly (ProcessGroup.allreduce) instead of through torch.distributed.all_reduce:
        # the Python wrapper's argument checks, rank lookups and logging hooks are ~10 us per call on the host, and an 8-pair step
        # issues seven of them from inside its backward pass, where the host is the bottleneck (one-rank RCCL step at 8 pairs 1.39 ms
        # against 1.30 plain; at 32 / 64 pairs the overhead is 2.2 / 1.6 %).
        # (the group object is looked up per call, not kept: a reference held here outlives destroy_process_group() and the backend's
        # threads are then torn down at interpreter exit -- "terminate called without an active exception", one gloo worker in three)
        self._sum = dist.AllreduceOptions()
        self._sum.reduceOp = dist.ReduceOp.SUM
        # ... and on the GPU, past torch altogether: RCCL's own entry point on the group's communicator (_NativeRccl).
        # COLVO_DDP_TORCH_COLLECTIVES=1 (or native_collectives=False) keeps the ProcessGroup calls; a torch build without _comm_ptr()
        # falls back with a warning.
        self._native = None
        self.native_fallback = None             # why the native path was asked for and is not in use (bench.py reports it)
        if (native_collectives and os.environ.get("COLVO_DDP_TORCH_COLLECTIVES", "0") in ("", "0")
                and dist.get_backend(process_group) == "nccl"
                and any(getattr(m, "flat_grad", None) is not None and m.flat_grad.is_cuda for m in modules)):
            try:
                self._native = _NativeRccl(process_group)
            except Exception as e:          # noqa: BLE001 -- the torch path is complete on its own
                import warnings
                self.native_fallback = f"{type(e).__name__}: {e}"
                warnings.warn(f"GradBuckets: native RCCL path unavailable ({type(e).__name__}: {e}); using ProcessGroup.allreduce",
                              RuntimeWarning, stacklevel=2)

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
        w = pg.allreduce([t], self._sum)
        if not (t.is_cuda and torch.cuda.is_current_stream_capturing()):
            self._eager_works.append(w)         # (see drain_eager_collectives; captured collectives never reach the watchdog's list)
            if len(self._eager_works) > 64:
                self._eager_works = [x for x in self._eager_works if not x.is_completed()]
        return w

    def drain_eager_collectives(self) -> None:
        """ProcessGroup.allreduce path: wait for every eager collective this object issued and check that each reports completion
        (graph.GraphedTrainStep's opt-in guard in front of a capture).  The native path has nothing to drain."""
        works, self._eager_works = self._eager_works, []
        for w in works:
            w.wait()
        if works and torch.cuda.is_available():
            torch.cuda.synchronize()
        for w in works:
            if not w.is_completed():
                raise RuntimeError("GradBuckets: an eager collective does not report completion after a device synchronisation")

    # ---- the spec's batch normalisation under data parallelism ------------------------------------ #
    def _install_reducer(self, on: bool) -> None:
        from . import functional
        if on and self.exact_batch_loss:
            functional.set_batch_reducer(self._reduce_loss_state)
        elif functional._batch_reducer == self._reduce_loss_state:
            functional.set_batch_reducer(None)

    def _reduce_loss_state(self, state: torch.Tensor, can_defer: bool = False):
        """state = {loss, 1/max(3n,1), n_valid, masked sum} of this rank's forward (device): -> the whole batch's (functional.py)."""
        from . import _lib
        if can_defer and self._defer:
            # Start the exchange and go on: the sum of the two floats travels beside the backward pass (RCCL's stream / the process
            # group's own), finish() waits for it.  The gradients' consumers get a scale of ONE in place of state[1].
            if self._deferred is not None:
                raise RuntimeError("GradBuckets(defer_loss_normalisation=True): a second photometric_loss call in one step (the first "
                                   "one's normaliser is still pending); use one loss call per step or leave the option off")
            dev = state.device
            if self._one is None or self._one.device != dev:
                self._one = torch.ones(1, device=dev)
                self._norm = torch.ones(1, device=dev)
            work = None
            if self._native is not None:
                self._native.all_reduce(state[2:4])
                if not torch.cuda.is_current_stream_capturing():
                    state.record_stream(self._native.stream)
            else:
                # a tensor of its own over the same two floats -- NOT a view of `state`: a process group that completes on a thread of
                # its own (gloo) bumps the version counter of what it reduced some time later, and autograd refuses the loss (a view of
                # `state`) if "another view of its base" changed in place meanwhile
                xch = torch.empty(0, device=dev, dtype=state.dtype).set_(state.untyped_storage(), state.storage_offset() + 2, (2,))
                work = self._all_reduce(xch)
            self._deferred = (state, work)
            return self._one
        # (also with one rank: `bench.py --rccl-single` then prices this collective like the bucket ones, and the state comes back
        # bit for bit -- a one-rank sum is the identity and the rescale repeats the forward's own arithmetic)
        if self._native is not None:
            self._native.all_reduce_now(state[2:4])
        else:
            self._all_reduce(state[2:4]).wait()                                       # (the caller's stream waits for it)
        _lib.check(_lib.load().colvo_warp_loss_rescale(_lib.ptr(state), self.world, _lib.stream_ptr()), "colvo_warp_loss_rescale")
        return None

    _one = None
    _norm = None

    def barrier(self) -> None:
        """All ranks here, this rank's device idle -- without a torch collective on the group when the native path is on."""
        if self._native is not None:
            self._native.barrier()
        else:
            dist.barrier(group=self.group)
            if torch.cuda.is_available():
                torch.cuda.synchronize()

    @property
    def native_collectives(self) -> bool:
        return self._native is not None

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
        buf = None
        if st.staging is not None:
            buf = st.staging[lo:hi]
            self._cast(sl, buf, True)    # conversion on the stream the hook runs on; the collective is ordered after it
        if self._native is not None:
            self._native.all_reduce(sl if buf is None else buf)
            self._pending.append((None, sl if buf is not None else None, buf))
        else:
            self._pending.append((self._all_reduce(sl if buf is None else buf), sl if buf is not None else None, buf))

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
        if self._native is not None:
            self._native.join()
        if self._deferred is not None:
            # the loss normaliser whose exchange was started behind the loss kernel: wait, turn the sums into the global loss and
            # scale, and post the scale to the optimizer (it multiplies it into grad_scale inside the update kernel, once)
            from . import _lib
            state, work = self._deferred
            self._deferred = None
            if work is not None:
                work.wait()
            _lib.check(_lib.load().colvo_warp_loss_rescale_to(_lib.ptr(state), self.world, _lib.ptr(self._norm), _lib.stream_ptr()),
                       "colvo_warp_loss_rescale_to")
            self._opt.grad_scale_dev = self._norm
        for work, sl, buf in self._pending:
            if work is not None:
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
        self._deferred = None                   # (an exchange started by a step that never reached finish())
        if getattr(self, "_shared_modules", None):
            from . import nn
            nn.unshare_side_stream(self._shared_modules)      # each network back on its own weight-gradient stream
            self._shared_modules = None
        if self._queue_claim is not None:
            self._queue_claim.release()
            self._queue_claim = None
