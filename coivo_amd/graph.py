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


class GraphedTrainStep:
    def __init__(self, depth_net, pose_net, optimizer, B: int, H: int, W: int, ddp=None, ssim_weight: float = 0.85,
                 warmup: int = 2, capture_policy: int = 2, capture_group: int = 2, full_loss: bool = False):
        dev = depth_net.flat_param.device
        self.depth_net, self.pose_net, self.opt, self.ddp = depth_net, pose_net, optimizer, ddp
        self.capture_policy, self.capture_group = int(capture_policy), int(capture_group)
        self.B, self.ssim_weight, self.full_loss = B, ssim_weight, bool(full_loss)
        self.frames = torch.zeros(2 * B, 3, H, W, device=dev)
        self.K = torch.zeros(B, 3, 3, device=dev)
        self.loss = torch.zeros((), device=dev)
        self._one = torch.ones((), device=dev)      # dL/dloss: a persistent scalar instead of a ones_like() fill per step
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self._warmup = warmup

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

    def capture(self) -> None:
        """Warm up (kernel attribute setup, side streams, allocator pools) without side effects, then capture."""
        nets = (self.depth_net, self.pose_net)
        self.opt.use_device_step_counter()          # a captured host-side step number would repeat at every replay
        snap_p = [n.flat_param.clone() for n in nets]
        snap_o = [{k: v.clone() for k, v in st.items()} for st in self.opt.state]
        side = torch.cuda.Stream(device=self.frames.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(self._warmup):
                self._step()
        torch.cuda.current_stream().wait_stream(side)
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
        from . import _lib
        _lib.check(_lib.load().colvo_set_capture_policy(self.capture_policy, self.capture_group), "colvo_set_capture_policy")
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = self._step()
            self.loss.copy_(out)
        self.graph = g
        restore()          # capture itself does not execute, but keep the state exactly as the caller left it
        torch.cuda.synchronize()

    def __call__(self, frames: Optional[torch.Tensor] = None, K: Optional[torch.Tensor] = None) -> torch.Tensor:
        if self.graph is None:
            self.capture()
        if frames is not None:
            self.frames.copy_(frames, non_blocking=True)
        if K is not None:
            self.K.copy_(K, non_blocking=True)
        self.graph.replay()
        # the captured Adam step rewrote the master weights: an eager forward afterwards must re-pack its operand copies
        self.depth_net.mark_params_changed()
        self.pose_net.mark_params_changed()
        return self.loss
