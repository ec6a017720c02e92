"""hipGraph-captured DCDP training step (BASELINE configs[4]: "hipGraph-captured train step and bucketed all-reduce").

Every entry point of libcolvo only enqueues on the given stream (no allocation, no sync, step counter and loss state on the
device), so the whole step -- zero-grad, DepthNet + PoseNet forward, fused loss, backward, RCCL buckets, fused Adam -- is
captured once (torch.cuda.CUDAGraph is the capture plumbing) and replayed as a single launch.

How the two-stream backward gets into the graph (round 3): NOT by capturing two streams.  While the stream is being captured,
`colvo_run_commands` builds the graph natively from the recorded command lists: the main-stream commands become one chain of
kernel nodes, the weight-gradient commands a second chain, tied together by explicit dependencies on the ONE capturing stream
(hipStreamGetCaptureInfo_v2 / hipStreamUpdateCaptureDependencies; csrc/program.hip).  The policy decides how many cross-branch
edges there are: ROCm 7.2 replays each branch on a stream of its own and pays for every edge between them -- round 2's
two-stream capture (one edge per layer) replayed at 2.99 ms against 1.52 ms eager, a single branch at 1.71 ms.
  capture_policy 0: one branch;  1: one edge per layer (the eager schedule);  2 (default): the weight gradients in segments of
  `capture_group` commands, each hanging off the main chain as captured so far.
The collectives of data parallelism (ddp.GradBuckets) are captured by stream capture as torch issues them (RCCL calls are
capturable), behind the weight-gradient chain joined at that point.

Inputs live in static device buffers (`frames` = [2B,3,H,W]: target frames then reference frames, `K`); the caller
writes the next batch into them (or passes tensors to __call__, which copies) and replays.
"""
from __future__ import annotations

from typing import Optional

import torch

from .functional import photometric_loss


GRAPH_STAT_NAMES = ("nodes", "edges", "roots", "leaves", "forks", "joins", "max_out_degree", "max_in_degree",
                    "calls", "main_commands", "side_commands", "side_segments", "chain_joins", "calls_with_carried_commands",
                    "max_entry_dependencies", "pending_commands")


class GraphedTrainStep:
    """carry (default on): the weight-gradient chain stays open from one recorded pass to the next inside the graph, as the eager
    schedule's deferred join leaves PoseNet's weight gradients running beside DepthNet's backward pass (include/colvo.h
    colvo_set_capture_carry); off: every pass ends joined (the round-3 form)."""

    def __init__(self, depth_net, pose_net, optimizer, B: int, H: int, W: int, ddp=None, ssim_weight: float = 0.85,
                 warmup: int = 2, capture_policy: int = 2, capture_group: Optional[int] = None, full_loss: bool = False,
                 carry: bool = True, allow_process_group_capture: bool = False):
        dev = depth_net.flat_param.device
        self.depth_net, self.pose_net, self.opt, self.ddp = depth_net, pose_net, optimizer, ddp
        # segments of the weight-gradient chain: one command each from 32 pairs on (the eager schedule node for node: replay 7.35
        # vs eager 7.35 ms at 64 pairs, 7.26 vs 7.11 with segments of two), two below (1.50 vs 1.55 ms at 8 pairs) -- a cross-chain
        # edge costs a fixed ~1-2 us, a delayed weight gradient costs in proportion to the kernels
        if capture_group is None:
            capture_group = 1 if B >= 32 else 2
        self.capture_policy, self.capture_group, self.carry = int(capture_policy), int(capture_group), bool(carry)
        self.B, self.ssim_weight, self.full_loss = B, ssim_weight, bool(full_loss)
        self.allow_process_group_capture = bool(allow_process_group_capture)
        self.frames = torch.zeros(2 * B, 3, H, W, device=dev)
        self.K = torch.zeros(B, 3, 3, device=dev)
        self.loss = torch.zeros((), device=dev)
        self._one = torch.ones((), device=dev)      # dL/dloss: a persistent scalar instead of a ones_like() fill per step
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.stats = None                           # colvo_graph_stats of the captured graph (GRAPH_STAT_NAMES)
        self._warmup = warmup
        self._spacers = []

    # one eager step on the static buffers
    def _step(self) -> torch.Tensor:
        B = self.B
        self.opt.zero_grad()
        if self.full_loss:      # the widened objective: two native calls, capture-safe (its workspace comes from the graph's pool)
            from .nn import dcdp_forward
            loss = dcdp_forward(self.depth_net, self.pose_net, None, None, self.K, ssim_weight=self.ssim_weight, full_loss=True,
                                frames=self.frames)[0]
        else:
            d_t, d_r, d_l = self.depth_net.forward_pair_split(self.frames)
            tgt, ref = self.frames[:B], self.frames[B:]
            pose, a, b = self.pose_net(tgt, ref, d_t, d_r)
            loss = photometric_loss(tgt, ref, d_l, pose, self.K, a, b, ssim_weight=self.ssim_weight)
        loss.backward(gradient=self._one)
        if self.ddp is not None:
            self.ddp.finish()
        self.opt.step()
        return loss.detach()

    def _operands_current(self) -> None:
        """The captured optimizer step (colvo_adam_pack_step) rewrites the operand copies of the weights itself, so the graph holds
        no repacking node: the copies the first forward kernels read must be current BEFORE a replay (round 3 captured
        two k_pack_weights_multi launches per step that the fused update made redundant).  No-op when they are."""
        if self.opt.writes_operand_copies():
            self.depth_net._prepare_weights()
            self.pose_net._prepare_weights()

    def capture(self) -> None:
        """Warm up (kernel attribute setup, side streams, allocator pools) without side effects, then capture."""
        from . import _lib
        lib = _lib.load()
        pg_path = self._process_group_path()        # (refuses a ProcessGroup.allreduce step the caller did not opt in for: before anything changes)
        nets = (self.depth_net, self.pose_net)
        for n in nets:
            n._grads_clean = False                  # whatever an earlier step left: the captured step starts with its clearing launch
        self.opt.use_device_step_counter()          # a captured host-side step number would repeat at every replay
        snap_p = [n.flat_param.clone() for n in nets]
        snap_o = [{k: v.clone() for k, v in st.items()} for st in self.opt.state]
        # Warm-up ON THE CALLER'S STREAM, not on a stream made for the purpose (round 4).  The runtime deals streams onto its (default
        # four) hardware queues in creation order: with a warm-up stream created first, the networks' weight-gradient streams and the
        # library's auxiliary stream -- created inside the first step -- wrapped around onto the queue of the stream the caller runs
        # eager steps on, and every eager step after a capture took 4.4 ms instead of 1.45 (tools/probe_eager_after_capture.py
        # --capture-first).  Created from the caller's stream they land where a plain eager loop puts them:
        # eager steps before, between and after replays run at their usual rate (tests/test_graph_gpu.py).
        for _ in range(self._warmup):
            self._step()
        torch.cuda.synchronize()

        def restore():
            with torch.no_grad():
                for n, p in zip(nets, snap_p):
                    n.flat_param.copy_(p)
                    n.mark_params_changed()
                for st, src in zip(self.opt.state, snap_o):
                    for k in st:
                        st[k].copy_(src[k])
        restore()
        # without the fused update the graph must hold the repacking nodes: leave the copies marked stale so that they are recorded
        self._operands_current()
        _lib.check(lib.colvo_set_capture_policy(self.capture_policy, self.capture_group), "colvo_set_capture_policy")
        _lib.check(lib.colvo_set_capture_carry(int(self.carry)), "colvo_set_capture_carry")
        _lib.check(lib.colvo_graph_stats_reset(), "colvo_graph_stats_reset")
        if pg_path:
            import time
            self.ddp.drain_eager_collectives()  # every work object of ours (the warm-up steps' too): waited for and completed
            torch.cuda.synchronize()
            time.sleep(2.5 * self._TORCH_WATCHDOG_PERIOD_S)
        g = torch.cuda.CUDAGraph()
        nspace = int(_lib.dev_env("COLVO_GRAPH_SPACER_STREAMS", "0"))       # (developer probe: streams created in front of the capture)
        self._spacers = [torch.cuda.Stream() for _ in range(nspace)]
        for sp in self._spacers:
            with torch.cuda.stream(sp):
                torch.zeros(1, device=self.frames.device)
        try:
            # thread_local: with a process group alive its watchdog thread polls the events of collectives still in flight; under
            # the default (global) capture mode that poll is "not permitted when stream is capturing", fails the capture and takes
            # the process down with it (seen at 64 pairs with --rccl-single; once in round 4's test runs)
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                out = self._step()
                self.loss.copy_(out)
                # every open chain is joined by now (the optimizer joins before it reads the gradients)
                import ctypes as C
                buf = (C.c_longlong * 16)()
                rc = lib.colvo_graph_stats(_lib.stream_ptr(), buf, 16)      # (the library's own counters are filled even if the
                self.stats = dict(zip(GRAPH_STAT_NAMES, [int(v) for v in buf]))   # runtime refuses to list the graph's nodes)
                if rc != 0:
                    self.stats["error"] = lib.colvo_last_error().decode("utf-8", "replace")
        except BaseException:
            # a failed capture leaves node handles of a graph that will never exist in the library's bookkeeping
            lib.colvo_capture_reset(None)
            raise
        finally:
            lib.colvo_set_capture_carry(0)
        if self.stats["pending_commands"]:
            lib.colvo_capture_reset(None)
            raise RuntimeError("hipGraph capture ended with weight-gradient commands still held back (a join is missing)")
        self.graph = g
        restore()          # capture itself does not execute, but keep the state exactly as the caller left it
        torch.cuda.synchronize()

    # torch's ProcessGroupNCCL watchdog wakes every kWatchdogThreadSleepMillis = 100 ms (a constant of ProcessGroupNCCL.hpp, not an option)
    _TORCH_WATCHDOG_PERIOD_S = 0.1

    def _process_group_path(self) -> bool:
        """The rule for a captured step that carries RCCL collectives (round 6).  -> True when the step's
        collectives go through ProcessGroup.allreduce AND the caller opted in (capture() then runs the timing-based guard).

        The hazard (two process aborts in round 5, gpurun_out/r5d/t_all.log and r5final/t_all.log:33-34): ProcessGroupNCCL keeps
        every EAGER collective on a list its watchdog thread walks, asking each one's end event whether it has completed.  Those
        events live on torch's internal RCCL stream, and this HIP runtime refuses hipEventQuery on an event of a stream that is
        being captured ('operation not permitted on an event last recorded in a capturing stream'); the refusal is an exception on
        the watchdog thread, and torch ends the process.  So the precondition of the abort is: torch's RCCL stream joins a capture
        while an eager work object is still listed.

        * Native path (GradBuckets(native_collectives=True), ddp._NativeRccl) -- the rule: the step's collectives, eager AND captured,
          are ncclAllReduce calls on a stream of OURS.  They create no work objects, and torch's RCCL stream never joins the capture,
          so whatever the caller left on the watchdog's list (a barrier at start-up, a logging reduction) is queried on a stream that
          is not capturing.  Nothing to wait for: the precondition cannot arise.
        * ProcessGroup.allreduce path: refused unless the caller opts in (allow_process_group_capture=True).  There is no API that says
          "the watchdog's list is empty" (no _wait_for_pending_works in this torch), so the opt-in is the round-5 guard made as tight
          as the API allows: every work object the library created is waited for and must report is_completed(), the device is
          drained, and one watchdog period (the constant above, twice over) passes before the capture begins; the caller must not
          issue a collective on the group until capture() returns.  Timing-based by nature -- hence not the default."""
        if self.ddp is None:
            return False
        try:
            import torch.distributed as dist
            nccl = bool(dist.is_initialized() and dist.get_backend(self.ddp.group) == "nccl")
        except Exception:           # noqa: BLE001
            nccl = False
        if not nccl or getattr(self.ddp, "native_collectives", False):
            return False
        if not self.allow_process_group_capture:
            raise RuntimeError(
                "GraphedTrainStep: a captured step on an nccl process group needs the native RCCL path -- GradBuckets(..., "
                "native_collectives=True) -- which creates no ProcessGroupNCCL work objects; through ProcessGroup.allreduce the watchdog "
                "thread can query an eager collective's event while RCCL's stream is capturing, and torch then aborts the process "
                "(coivo_amd/graph.py _process_group_path).  allow_process_group_capture=True accepts the timing-based guard.")
        return True

    def close(self) -> None:
        """Destroy the captured graph NOW, at a defined point, and return the library to its pre-capture state.

        Without it the teardown happens whenever the last reference goes: torch's CUDAGraph destructor calls hipGraphExecDestroy /
        hipGraphDestroy, gives the graph's private pool back and synchronises the device (ROCm >= 6.2 defers the release of a graph
        exec's resources to the next synchronising call), the library keeps the dead graph's node handles and held-back command
        copies until the next capture, and the networks -- which sit in reference cycles (recorded passes, autograd closures) --
        keep their persistent buffers until a full garbage collection.  A process that re-captures (another batch size, another
        policy; round 4's test run: eight captures in one process) then tears graphs down in the middle of unrelated work.  Here:
        device idle, nothing capturing, graph exec and graph destroyed, pool released, library reset (colvo_capture_reset), the
        capture-only scratch of the step dropped.  The step object can capture() again afterwards.  DESIGN.md section 3.4."""
        import gc
        from . import _lib
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("GraphedTrainStep.close() while a stream is being captured")
        torch.cuda.synchronize()
        g, self.graph = self.graph, None
        if g is not None:
            g.reset()                   # hipGraphExecDestroy + hipGraphDestroy + the private pool, while the device is idle
            del g
        self._spacers = []
        self.stats = None
        _lib.check(_lib.load().colvo_capture_reset(_lib.stream_ptr()), "colvo_capture_reset")
        for n in (self.depth_net, self.pose_net):
            n.join_side()
        gc.collect()                    # cyclic garbage of earlier steps' networks goes now, not inside somebody else's capture
        torch.cuda.synchronize()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __call__(self, frames: Optional[torch.Tensor] = None, K: Optional[torch.Tensor] = None) -> torch.Tensor:
        if self.graph is None:
            self.capture()
        if frames is not None:
            self.frames.copy_(frames, non_blocking=True)
        if K is not None:
            self.K.copy_(K, non_blocking=True)
        self._operands_current()
        self.graph.replay()
        clean = bool(getattr(self.opt, "zero_grad_in_step", False)) and self.opt.writes_operand_copies()
        for n in (self.depth_net, self.pose_net):
            n._grads_clean = clean           # the replayed update cleared the arenas (or did not): an eager zero_grad() may skip
            if self.opt.writes_operand_copies():
                n.operands_written()         # the captured update left master weights AND operand copies current
            else:
                n.mark_params_changed()      # the captured Adam step rewrote the master weights: an eager forward must re-pack
        return self.loss
