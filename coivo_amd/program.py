"""Recorded command lists: the host side of colvo_run_commands (include/colvo.h).

`ops.*` calls made while a Program is recording are appended to it instead of being launched.  The recorded list is then
replayed with ONE C-ABI call per network pass.  All tensors a command refers to are kept alive by the Program (the
activation buffers of a plan are therefore persistent); the few pointers that differ from call to call -- input images,
the output, incoming gradients -- are registered as externals and patched before each replay.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib

_active: Optional["Program"] = None


def recording() -> Optional["Program"]:
    return _active


class Program:
    def __init__(self):
        self.cmds: List[_lib.Cmd] = []
        self.keep: List[object] = []
        self.stream = 0                      # stream selector of the commands being recorded (0 main, 1 side)
        self.marks: List[Tuple[int, object]] = []     # (number of commands recorded so far, payload) -- hook points
        self._ext: Dict[str, int] = {}       # external name -> data pointer at record time
        self._slots: Dict[str, List[Tuple[int, int]]] = {}
        self._arr = None
        self.uses_side = False
        self.flag_slots: List[Tuple[int, int]] = []   # (command, integer slot) of per-replay flags (set_flags)
        self._flag_value = 0

    # ---- recording ------------------------------------------------------------------------------------------- #
    def __enter__(self):
        global _active
        if _active is not None:
            raise RuntimeError("nested Program recording")
        _active = self
        return self

    def __exit__(self, *exc):
        global _active
        _active = None
        if exc[0] is None:
            self._finalize()
        return False

    def external(self, name: str, t: torch.Tensor) -> torch.Tensor:
        """Declare `t` (used by the commands recorded from now on) as a per-call pointer called `name`."""
        self._ext[name] = t.data_ptr()
        return t

    def add(self, op: int, desc=None, p: Sequence[Optional[torch.Tensor]] = (), i: Sequence[int] = (),
            f: Sequence[float] = (), raw: Sequence = (), flag_slot: Optional[int] = None) -> None:
        """raw: [(slot, address, keepalive)] -- pointer slots that are not tensors (a host-side table the command refers to)."""
        c = _lib.Cmd()
        c.op, c.stream = op, self.stream
        if desc is not None:
            c.desc = desc
        for slot, address, keepalive in raw:
            c.p[slot] = address
            self.keep.append(keepalive)
        for k, t in enumerate(p):
            if t is not None:
                c.p[k] = t.data_ptr()
                self.keep.append(t)
        for k, v in enumerate(i):
            c.i[k] = int(v)
        for k, v in enumerate(f):
            c.f[k] = float(v)
        if self.stream:
            self.uses_side = True
        if flag_slot is not None:            # an integer argument the host sets before each replay (set_flags)
            self.flag_slots.append((len(self.cmds), flag_slot))
        self.cmds.append(c)

    def fork(self) -> None:
        self.uses_side = True
        self.add(_lib.CMD_FORK)

    def join(self) -> None:
        self.add(_lib.CMD_JOIN)

    def side_sync(self) -> None:
        """(recorded as a side command) the side stream in use waits for every other side stream's work so far."""
        self.add(_lib.CMD_SIDE_SYNC)

    def mark(self, payload) -> None:
        self.marks.append((len(self.cmds), payload))

    def _finalize(self) -> None:
        n = len(self.cmds)
        self._arr = (_lib.Cmd * max(n, 1))(*self.cmds)
        by_ptr = {ptr: name for name, ptr in self._ext.items()}
        if len(by_ptr) != len(self._ext):
            raise RuntimeError("two externals of a Program share one address")
        self._slots = {name: [] for name in self._ext}
        for ci in range(n):
            for k in range(_lib.NPTR):
                name = by_ptr.get(self._arr[ci].p[k])
                if name is not None:
                    self._slots[name].append((ci, k))
        self.cmds = []   # the array is the program now

    # ---- replay ---------------------------------------------------------------------------------------------- #
    def patch(self, name: str, t: torch.Tensor) -> None:
        ptr = t.data_ptr()
        arr = self._arr
        for ci, k in self._slots[name]:
            arr[ci].p[k] = ptr

    def set_flags(self, value: int) -> None:
        """Write `value` into every registered per-replay integer slot (today: "the gradient arena is still zero" of the weight-
        gradient commands, include/colvo.h COLVO_CMD_CONV_WGRAD)."""
        value = int(value)
        if value == self._flag_value or self._arr is None:
            if self._arr is None:
                for ci, k in self.flag_slots:
                    self.cmds[ci].i[k] = value
                self._flag_value = value
            return
        arr = self._arr
        for ci, k in self.flag_slots:
            arr[ci].i[k] = value
        self._flag_value = value

    def run(self, side: Optional[torch.cuda.Stream], begin: int = 0, end: Optional[int] = None) -> None:
        n = len(self._arr) if end is None else end
        if n <= begin:
            return
        lib = _lib.load()
        base = C.addressof(self._arr) + begin * C.sizeof(_lib.Cmd)
        _lib.check(lib.colvo_run_commands(base, n - begin, _lib.stream_ptr(),
                                          0 if side is None else side.cuda_stream), "colvo_run_commands")

    def __len__(self) -> int:
        return len(self._arr) if self._arr is not None else len(self.cmds)
