"""hipcc driver: compiles coivo_amd/csrc/*.hip for gfx950 into coivo_amd/lib/libcolvo.so (in-tree).

    python -m coivo_amd.build [--force]

The .so is git-ignored (history stays source-only) but travels to the GPU box with the tree.
"""
from __future__ import annotations

import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "lib", "obj")
LIB = os.path.join(LIBDIR, "libcolvo.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
         "-fno-gpu-rdc", "-DNDEBUG",
         # CDNA4 executes packed f32 (v_pk_*_f32) at the plain-op rate, so SLP packing only adds the
         # v_mov pairs that feed it (measured: fused-loss fwd 112 -> 47 VGPRs, -21 % VALU without it)
         "-fno-slp-vectorize"]
# per-file extras.  conv.hip: MFMA accumulators in VGPRs, never AGPRs -- its mfma_result_guard() ties them to an asm
# statement with "+v", and an AGPR accumulator would be copied out (= read) in front of that statement
# (profiles/r2_mfma_hazard.md).  wgrad.hip keeps AGPR accumulators (faster pixel loop) and ties them with "+a".
FILE_FLAGS = {"conv.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"], "bwd16.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
              "fwd16.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"], "conv_rt.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
              "wgrad_rt.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def _deps():
    return glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "colvo.h")]


def _stale(out, srcs):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(s) > t for s in srcs)


def _object_stamp(src) -> str:
    """sha256 of everything one object is built from: its source, every header it may include, its flags.  Kept in a
    sidecar file beside the object; an object is reused only when its stamp matches (mtimes do not survive a copy of the
    tree, and a cached object NEWER than an edited source must not be linked)."""
    import hashlib
    h = hashlib.sha256()
    for f in [src] + sorted(_deps()):
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update((" ".join(FLAGS) + " " + " ".join(FILE_FLAGS.get(os.path.basename(src), []))).encode())
    return h.hexdigest()


def _compile(src):
    obj = os.path.join(OBJDIR, os.path.basename(src)[:-4] + ".o")
    stamp_file, stamp = obj + ".stamp", _object_stamp(src)
    try:
        current = os.path.exists(obj) and open(stamp_file).read().strip() == stamp
    except OSError:
        current = False
    if not current:
        if os.path.exists(stamp_file):
            os.remove(stamp_file)
        cmd = [HIPCC] + FLAGS + FILE_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
        with open(stamp_file, "w") as f:
            f.write(stamp)
    return obj


HASHFILE = os.path.join(LIBDIR, "libcolvo.srchash")


def source_hash() -> str:
    """sha256 over every source the library is built from (csrc/*.hip, csrc/*.h, include/colvo.h) and the flags.
    Stored beside the .so by build(); _lib.load() refuses a library whose hash differs from the tree's (the .so is
    git-ignored and travels with the tree, so an edited kernel must never run against a stale binary)."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(CSRC, "*.hip"))) + sorted(_deps())
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update((" ".join(FLAGS) + repr(sorted(FILE_FLAGS.items()))).encode())
    return h.hexdigest()


def is_current() -> bool:
    try:
        with open(HASHFILE) as f:
            return os.path.exists(LIB) and f.read().strip() == source_hash()
    except OSError:
        return False


def ensure() -> str:
    """Build unless the in-tree library was built from exactly these sources."""
    return LIB if is_current() else build()


def emit_asm(src_name: str, out: str) -> str:
    """gfx950 ISA listing of one csrc file with the library's flags (tools/isa_check_mfma.py, tests/test_isa_cpu.py)."""
    src = os.path.join(CSRC, src_name)
    cmd = [HIPCC] + [f for f in FLAGS if f != "-fPIC"] + FILE_FLAGS.get(src_name, []) + ["--offload-device-only", "-S", src, "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc -S failed for {src}:\n{r.stdout}\n{r.stderr}")
    return out


def build(force: bool = False) -> str:
    os.makedirs(OBJDIR, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    if not srcs:
        raise RuntimeError("no HIP sources found")
    if force:
        for f in glob.glob(os.path.join(OBJDIR, "*.o")) + glob.glob(os.path.join(OBJDIR, "*.stamp")):
            os.remove(f)
    # every object carries a sidecar stamp = hash of its own inputs (_object_stamp): edited sources, headers or flags recompile
    # exactly the objects they affect, whatever the files' mtimes say
    for f in glob.glob(os.path.join(OBJDIR, "*.o")):
        if os.path.basename(f)[:-2] + ".hip" not in {os.path.basename(x) for x in srcs}:
            os.remove(f)            # object of a source file that no longer exists
    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        objs = list(ex.map(_compile, srcs))
    if force or _stale(LIB, objs) or not is_current():       # (a relink is cheap; the objects above are already exact)
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        with open(HASHFILE, "w") as f:
            f.write(source_hash())
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
