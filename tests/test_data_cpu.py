"""CPU tests of the input pipeline's host logic: folder scan, pair enumeration, sharding, intrinsics."""
import os

import numpy as np
import pytest
import torch

from coivo_amd import data as D
from oracle import colvo_spec as S


def make_tree(root, seqs=(("a", 5, (48, 64)), ("b", 3, (48, 64))), cam=("a",), fmt="png", seed=0):
    from PIL import Image
    rng = np.random.default_rng(seed)
    for name, n, (h, w) in seqs:
        d = os.path.join(root, name)
        os.makedirs(d)
        for k in range(n):
            a = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
            if fmt == "npy":
                np.save(os.path.join(d, f"{k:06d}.npy"), a)
            else:
                Image.fromarray(a).save(os.path.join(d, f"{k:06d}.{fmt}"))
        if name in cam:
            np.savetxt(os.path.join(d, "cam.txt"), np.array([[50.0, 0, 31.0], [0, 52.0, 23.0], [0, 0, 1]]))
    return root


def test_folder_scan_and_pairs(tmp_path):
    ds = D.SequenceFolder(make_tree(str(tmp_path)))
    assert len(ds) == 4 + 2
    it = ds[0]
    assert it["sequence"] == "a" and it["index"] == 0 and it["tgt"].shape == (48, 64, 3) and it["tgt"].dtype == np.uint8
    assert torch.equal(it["K"], torch.tensor([[50.0, 0, 31.0], [0, 52.0, 23.0], [0, 0, 1]]))
    last = ds[5]
    assert last["sequence"] == "b" and last["index"] == 1
    assert torch.equal(last["K"], D.default_intrinsics(48, 64))
    # consecutive samples of a sequence share a frame: ref of pair k is tgt of pair k+1
    assert np.array_equal(ds[0]["ref"], ds[1]["tgt"])
    ds2 = D.SequenceFolder(str(tmp_path), skip=2)
    assert len(ds2) == 3 + 1 and np.array_equal(ds2[0]["ref"], ds[1]["ref"])
    with pytest.raises(ValueError):
        D.SequenceFolder(str(tmp_path), skip=7)
    with pytest.raises(FileNotFoundError):
        D.SequenceFolder(str(tmp_path / "missing"))


def test_npy_frames_and_bad_frame(tmp_path):
    ds = D.SequenceFolder(make_tree(str(tmp_path), fmt="npy"))
    assert ds[0]["tgt"].shape == (48, 64, 3)
    np.save(os.path.join(str(tmp_path), "a", "000000.npy"), np.zeros((48, 64), dtype=np.float32))
    with pytest.raises(ValueError):
        ds[0]


@pytest.mark.parametrize("n,batch,world", [(100, 4, 2), (37, 3, 4), (8, 8, 1), (7, 8, 1)])
def test_shards_are_disjoint_equal_and_cover(n, batch, world):
    shards = [D.shard_indices(n, batch, r, world, shuffle=True, seed=5, epoch=2) for r in range(world)]
    usable = (n // (world * batch)) * world * batch
    assert all(len(s) == usable // world for s in shards)
    flat = [i for s in shards for i in s]
    assert len(set(flat)) == len(flat) == usable and all(0 <= i < n for i in flat)
    again = D.shard_indices(n, batch, 0, world, shuffle=True, seed=5, epoch=2)
    assert again == shards[0]
    if usable:
        assert D.shard_indices(n, batch, 0, world, shuffle=True, seed=5, epoch=3) != shards[0]
    assert D.shard_indices(n, batch, 0, world, shuffle=False, seed=0, epoch=0) == list(range(0, usable, world))
    with pytest.raises(ValueError):
        D.shard_indices(n, batch, world, world, shuffle=False, seed=0, epoch=0)


def test_intrinsics_resize_matches_spec_and_geometry():
    K = torch.tensor([[50.0, 0, 31.0], [0, 52.0, 23.0], [0, 0, 1]])
    got = D.resize_intrinsics(K, (48, 64), (96, 160))
    assert torch.equal(got, S.resize_intrinsics(K, (48, 64), (96, 160)))
    assert got[0, 0].item() == 125.0 and got[1, 1].item() == 104.0
    # the source pixel's centre (x + 1/2) maps to (x + 1/2) * s in the half-pixel convention
    assert abs(got[0, 2].item() - ((31.0 + 0.5) * 2.5 - 0.5)) < 1e-6
    assert torch.equal(D.resize_intrinsics(K, (48, 64), (48, 64)), K)


def test_spec_resize_known_answers():
    u8 = torch.full((1, 4, 6, 3), 255, dtype=torch.uint8)
    assert torch.equal(S.resize_frames_u8(u8, 8, 12), torch.ones(1, 3, 8, 12))
    g = torch.Generator().manual_seed(1)
    u8 = torch.randint(0, 256, (2, 5, 7, 3), generator=g, dtype=torch.uint8)
    same = S.resize_frames_u8(u8, 5, 7)
    assert torch.equal(same, u8.permute(0, 3, 1, 2).float() / 255.0)
    # exact 2x downscale with half-pixel centres = the mean of each 2x2 block
    u8 = torch.randint(0, 256, (1, 8, 8, 3), generator=g, dtype=torch.uint8)
    half = S.resize_frames_u8(u8, 4, 4)
    want = u8.permute(0, 3, 1, 2).float().view(1, 3, 4, 2, 4, 2).mean(dim=(3, 5)) / 255.0
    assert (half - want).abs().max().item() < 1e-6


def test_loader_refuses_cpu(tmp_path):
    ds = D.SequenceFolder(make_tree(str(tmp_path)))
    with pytest.raises(ValueError):
        D.PairLoader(ds, 2, (50, 64))
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            D.PairLoader(ds, 2, (64, 96), device="cpu")


def test_native_npy_frame_reader_matches_numpy_and_rejects_bad_files(tmp_path):
    """colvo_read_npy_u8_frames (host-only entry point of the C-ABI: no GPU needed): payloads equal np.load, every header is
    checked against the expected frame shape, and a failing file is named in the error."""
    from coivo_amd.data import read_npy_frames
    rng = np.random.default_rng(3)
    h, w = 37, 52
    paths, ref = [], []
    for k in range(11):
        a = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        p = str(tmp_path / f"f{k:02d}.npy")
        np.save(p, a)
        paths.append(p)
        ref.append(a)
    for threads in (1, 4, 32):
        out = np.full((len(paths), h, w, 3), 7, dtype=np.uint8)
        read_npy_frames(paths, out, threads)
        assert np.array_equal(out, np.stack(ref))
    # version-2 header
    p2 = str(tmp_path / "v2.npy")
    with open(p2, "wb") as f:
        np.lib.format.write_array(f, ref[0], version=(2, 0))
    out = np.zeros((1, h, w, 3), dtype=np.uint8)
    read_npy_frames([p2], out)
    assert np.array_equal(out[0], ref[0])

    def expect_fail(path, what):
        with pytest.raises(RuntimeError, match=what):
            read_npy_frames([paths[0], path], np.zeros((2, h, w, 3), dtype=np.uint8), 2)

    bad = str(tmp_path / "shape.npy")
    np.save(bad, np.zeros((h, w + 1, 3), dtype=np.uint8))
    expect_fail(bad, "shape.npy")
    bad = str(tmp_path / "dtype.npy")
    np.save(bad, np.zeros((h, w, 3), dtype=np.float32))
    expect_fail(bad, "dtype.npy")
    bad = str(tmp_path / "fortran.npy")
    np.save(bad, np.asfortranarray(np.zeros((h, w, 3), dtype=np.uint8)))
    expect_fail(bad, "fortran.npy")
    bad = str(tmp_path / "short.npy")
    with open(paths[1], "rb") as f:
        blob = f.read()
    with open(bad, "wb") as f:
        f.write(blob[:-100])
    expect_fail(bad, "truncated")
    expect_fail(str(tmp_path / "missing.npy"), "missing.npy")
    bad = str(tmp_path / "junk.npy")
    with open(bad, "wb") as f:
        f.write(b"not a numpy file at all")
    expect_fail(bad, "junk.npy")
    with pytest.raises(ValueError):
        read_npy_frames(paths[:2], np.zeros((3, h, w, 3), dtype=np.uint8))


def test_decode_worker_process_protocol(tmp_path):
    """coivo_amd/_decode_worker.py (the image-decoding worker of PairLoader(decoders=N)) on its own: frames land in the shared-memory
    block at the requested offset, a wrong size or a missing file is reported, and the worker exits when stdin closes."""
    import subprocess
    import sys
    from multiprocessing import shared_memory
    from PIL import Image
    rng = np.random.default_rng(5)
    h, w = 24, 40
    frames = [rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8) for _ in range(3)]
    paths = []
    for k, a in enumerate(frames):
        p = str(tmp_path / f"{k}.png")
        Image.fromarray(a).save(p)
        paths.append(p)
    worker = os.path.join(os.path.dirname(D.__file__), "_decode_worker.py")
    shm = shared_memory.SharedMemory(create=True, size=3 * h * w * 3)
    proc = subprocess.Popen([sys.executable, worker], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, bufsize=1)
    try:
        def ask(off, hh, ww, path):
            proc.stdin.write(f"{shm.name}\t{off}\t{hh}\t{ww}\t{path}\n")
            proc.stdin.flush()
            return proc.stdout.readline().rstrip("\n")
        for k in (2, 0, 1):
            assert ask(k * h * w * 3, h, w, paths[k]) == "ok"
        got = np.ndarray((3, h, w, 3), dtype=np.uint8, buffer=shm.buf)
        assert np.array_equal(got, np.stack(frames))
        del got
        assert ask(0, h + 1, w, paths[0]).startswith("err ValueError")
        assert ask(0, h, w, str(tmp_path / "missing.png")).startswith("err FileNotFoundError")
        assert ask(0, h, w, paths[1]) == "ok"                   # still alive after errors
        proc.stdin.close()
        assert proc.wait(timeout=20) == 0
    finally:
        if proc.poll() is None:
            proc.kill()
        shm.close()
        shm.unlink()
