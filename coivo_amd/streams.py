"""Stream / hardware-queue policy of the training step: ONE place that decides how many library-owned side streams
`colvo_run_commands` may use, from who else drives hardware queues in this process.

Why a policy at all (measured on MI355X, ROCm 7.2): the step runs its input-gradient chain
on the caller's stream and the weight gradients on one or two side streams.  Beyond a small number of ACTIVE hardware queues
with cross-queue dependencies between them the runtime serialises the backward pass (1.3 ms -> 3.5-5 ms per step).  Rounds 2-3 took
that number to be three; round 4 measured it: FOUR queues are served at a time -- main + one side stream per network (two) + the
library's auxiliary stream is already four, which is why an external party's stream (RCCL's communicator stream, ddp.GradBuckets; a
loader's copy stream, data.PairLoader) has to displace the auxiliary stream -- unless the networks share ONE side stream
(nn.share_side_stream, what GradBuckets arranges), which leaves room for one external queue beside it (_free_claims below).

    claim = streams.claim_external_queue("rccl")   # the auxiliary stream is switched off while any claim is held
    ...
    claim.release()                                # the last release switches it back on

Two environment facts are read once (they bind at HIP initialisation, before this module can change them):
  * GPU_MAX_HW_QUEUES <= 2 folds every stream onto two hardware queues: extra streams are harmless and claims change nothing;
  * data parallel (WORLD_SIZE > 1) wants GPU_MAX_HW_QUEUES >= 8 (or <= 2), otherwise the side stream lands on the main stream's
    queue once the communicator exists: `check_environment()` warns (the step then runs ~15 % slower; nothing breaks).
"""
from __future__ import annotations

import os
import threading
from typing import Dict, Optional

_DEFAULT_AUX = 3            # library default: up to MAX_AUX, the effective number is min(COLVO_SIDE_STREAMS - 1, this)
_lock = threading.RLock()      # re-entrant: a QueueClaim collected while the lock is held releases itself
_claims: Dict[int, str] = {}
_next_id = 0
_base_aux = _DEFAULT_AUX    # what configure() asked for when no external queue is claimed
_shared_side = False        # the networks run their weight gradients on ONE shared side stream (nn.share_side_stream)


def hw_queue_limit(env=None) -> Optional[int]:
    """GPU_MAX_HW_QUEUES as the HIP runtime read it at initialisation (None: unset = the runtime's default of 4)."""
    v = (os.environ if env is None else env).get("GPU_MAX_HW_QUEUES")
    try:
        return int(v) if v is not None else None
    except ValueError:
        return None


def folded(env=None) -> bool:
    """True when the runtime folds all streams onto <= 2 hardware queues (extra streams then cost nothing)."""
    q = hw_queue_limit(env)
    return q is not None and q <= 2


def _free_claims() -> int:
    """External queues that fit beside the auxiliary stream.  Round 4 measured the real budget: FOUR queues are served at a time
    (main + DepthNet's side stream + PoseNet's side stream + the auxiliary stream run at full speed, a fifth stream costs 1.34 ->
    3.5 ms per step whatever GPU_MAX_HW_QUEUES says).  With one side stream per network nothing is left for an external party; when
    the networks SHARE one side stream (ddp.GradBuckets arranges that) one external queue -- RCCL's -- fits: the one-rank RCCL step
    measured 1.39 ms against 1.55 with the auxiliary stream switched off (and 3.67 with five streams)."""
    return 1 if _shared_side else 0


def _apply() -> int:
    n = _base_aux if (len(_claims) <= _free_claims() or folded()) else 0
    from . import _lib
    _lib.check(_lib.load().colvo_set_aux_side_streams(n), "colvo_set_aux_side_streams")
    return n


def configure(n_external_queues: int = 0, aux_side_streams: int = _DEFAULT_AUX) -> int:
    """Explicit form for hosts that drive streams of their own without going through GradBuckets / PairLoader: declare how
    many hardware queues OTHER than the step's main and side stream the process keeps busy.  Returns the number of auxiliary
    side streams colvo_run_commands may use from now on.  Replaces every earlier claim."""
    global _base_aux, _next_id, _shared_side
    if n_external_queues < 0 or aux_side_streams < 0:
        raise ValueError("streams.configure: counts must be >= 0")
    with _lock:
        _claims.clear()
        _shared_side = False
        _base_aux = aux_side_streams
        for _ in range(n_external_queues):
            _claims[_next_id] = "configured"
            _next_id += 1
        return _apply()


class QueueClaim:
    """One external hardware queue in use; release() (idempotent, also on garbage collection) gives it back."""

    def __init__(self, cid: int, who: str):
        self._cid, self.who = cid, who

    def release(self) -> None:
        cid, self._cid = self._cid, None
        if cid is None:
            return
        with _lock:
            if _claims.pop(cid, None) is not None:
                try:
                    _apply()
                except Exception:           # noqa: BLE001 -- interpreter shutdown: the library may be gone
                    pass

    def __del__(self):
        self.release()


def claim_external_queue(who: str) -> QueueClaim:
    global _next_id
    with _lock:
        cid = _next_id
        _next_id += 1
        _claims[cid] = who
        _apply()
    return QueueClaim(cid, who)


def external_queues() -> int:
    with _lock:
        return len(_claims)


def aux_side_streams() -> int:
    """The limit currently handed to the library."""
    with _lock:
        return _base_aux if (len(_claims) <= _free_claims() or folded()) else 0


def networks_share_side_stream(shared: bool) -> int:
    """Told by nn.share_side_stream(): the networks' weight gradients run on one common side stream (one hardware queue instead of
    one per network).  Returns the auxiliary-stream limit now in force."""
    global _shared_side
    with _lock:
        _shared_side = bool(shared)
        return _apply()


def reset() -> None:
    """Drop every claim and restore the library default (tests)."""
    configure(0, _DEFAULT_AUX)


def check_environment(world_size: int, env=None) -> Optional[str]:
    """Warn (once per message) when this process is a data-parallel rank whose environment promises the HIP runtime fewer than 8
    hardware queues; returns the warning text, or None when the setting is fine.  Fine: GPU_MAX_HW_QUEUES >= 8 (main, weight-gradient
    side stream and RCCL's stream each get a queue of their own) or <= 2 (every stream folded onto two queues: measured level with
    the three-queue schedule).  In between -- the runtime's default of 4 included -- the side stream lands on the main stream's
    queue once the communicator exists and the backward pass serialises (1.96 ms per step against 1.68).

    A warning, not an error: the setting costs ~15 %, it does not break anything; what this function can see is os.environ, not what
    the runtime bound when HIP initialised (a value exported after that passes falsely, one exported by a parent launcher and removed
    fails falsely); and the multi-rank RCCL path has not run on hardware yet -- a hard stop there would be the first thing an 8-GPU
    run hits.  COLVO_IGNORE_HW_QUEUES=1 silences it."""
    if world_size <= 1:
        return None
    e = os.environ if env is None else env
    q = hw_queue_limit(e)
    if (q is not None and (q >= 8 or q <= 2)) or e.get("COLVO_IGNORE_HW_QUEUES"):
        return None
    msg = (f"data parallel (world size {world_size}): GPU_MAX_HW_QUEUES is {'unset (= 4)' if q is None else q} in this process's "
           "environment; export GPU_MAX_HW_QUEUES=8 (or <= 2) BEFORE the process initialises HIP, as bench.py does -- with 3..7 the "
           "weight-gradient side stream shares a hardware queue with the main stream once the RCCL communicator exists and the "
           "backward pass serialises (measured 1.96 ms per step against 1.68)")
    import warnings
    warnings.warn(msg, RuntimeWarning, stacklevel=3)
    return msg
