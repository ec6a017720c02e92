"""coivo_amd/build.py: an object is reused only when the stamp beside it equals the hash of its own inputs (a cached
object NEWER than an edited source must not be linked -- mtimes do not survive a copy of the tree)."""
import os

import pytest


def test_object_stamp_decides_recompilation(tmp_path, monkeypatch):
    from coivo_amd import build
    src = os.path.join(build.CSRC, "api.hip")
    monkeypatch.setattr(build, "OBJDIR", str(tmp_path))
    calls = []

    class R:
        returncode, stdout, stderr = 0, "", ""

    def fake_run(cmd, **kw):
        calls.append(cmd)
        open(cmd[cmd.index("-o") + 1], "wb").write(b"obj")
        return R()
    monkeypatch.setattr(build.subprocess, "run", fake_run)
    obj = build._compile(src)
    assert len(calls) == 1 and os.path.exists(obj) and open(obj + ".stamp").read() == build._object_stamp(src)
    build._compile(src)
    assert len(calls) == 1                                      # stamp matches: reused
    os.utime(obj, (1, 1))                                       # an OLD object with a matching stamp is still current ...
    build._compile(src)
    assert len(calls) == 1
    open(obj + ".stamp", "w").write("0" * 64)                   # ... a NEW object whose stamp does not match is not
    build._compile(src)
    assert len(calls) == 2
    os.remove(obj + ".stamp")
    build._compile(src)
    assert len(calls) == 3


def test_stamp_covers_headers_and_flags(monkeypatch):
    from coivo_amd import build
    src = os.path.join(build.CSRC, "conv.hip")
    base = build._object_stamp(src)
    monkeypatch.setattr(build, "FLAGS", build.FLAGS + ["-DX"])
    assert build._object_stamp(src) != base
    monkeypatch.undo()
    monkeypatch.setitem(build.FILE_FLAGS, "conv.hip", [])
    assert build._object_stamp(src) != base
    assert build._object_stamp(os.path.join(build.CSRC, "wgrad.hip")) != base
