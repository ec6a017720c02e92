"""-m gpu parity of the input pipeline: the loader's batches against the oracle's resize of the same files."""
import numpy as np
import pytest
import torch

from tests.gpu_util import dev
from tests.test_data_cpu import make_tree

pytestmark = pytest.mark.gpu

PIXEL_TOL = 2e-6     # fp32 bilinear weights; values in [0,1]


def _convert(u8, H, W):
    from coivo_amd import _lib
    lib = _lib.load()
    n, h, w, _ = u8.shape
    out = torch.empty(n, 3, H, W, device=dev())
    _lib.check(lib.colvo_frames_u8_to_f32(_lib.ptr(u8), n, h, w, H, W, _lib.ptr(out), _lib.stream_ptr()), "frames")
    return out


@pytest.mark.parametrize("h,w,H,W", [(48, 64, 48, 64), (48, 64, 96, 160), (100, 130, 64, 96), (270, 350, 256, 320),
                                     (7, 5, 32, 32), (1080, 1350, 256, 320)])
def test_frames_kernel_matches_oracle(h, w, H, W):
    from oracle import colvo_spec as S
    g = torch.Generator().manual_seed(h * 7 + W)
    u8 = torch.randint(0, 256, (3, h, w, 3), generator=g, dtype=torch.uint8)
    want = S.resize_frames_u8(u8, H, W)
    got = _convert(u8.to(dev()), H, W).cpu()
    assert got.shape == want.shape
    assert (got - want).abs().max().item() < PIXEL_TOL
    if (h, w) == (H, W):
        assert torch.equal(got, want)


def test_loader_batches_match_oracle(tmp_path):
    from coivo_amd import data as D
    from oracle import colvo_spec as S
    ds = D.SequenceFolder(make_tree(str(tmp_path), seqs=(("a", 9, (60, 80)), ("b", 6, (60, 80)))))
    H, W = 64, 96
    seen = []
    for rank in range(2):
        ld = D.PairLoader(ds, 3, (H, W), rank=rank, world_size=2, shuffle=True, seed=3, workers=2)
        ld.set_epoch(1)
        idx = D.shard_indices(len(ds), 3, rank, 2, shuffle=True, seed=3, epoch=1)
        assert len(ld) == len(idx) // 3 == 2
        n = 0
        for step, batch in enumerate(ld):
            ids = idx[step * 3:(step + 1) * 3]
            items = [ds[i] for i in ids]
            tgt = S.resize_frames_u8(torch.from_numpy(np.stack([it["tgt"] for it in items])), H, W)
            ref = S.resize_frames_u8(torch.from_numpy(np.stack([it["ref"] for it in items])), H, W)
            K = torch.stack([S.resize_intrinsics(it["K"], (60, 80), (H, W)) for it in items])
            assert batch["tgt"].shape == (3, 3, H, W) and batch["tgt"].is_cuda
            assert (batch["tgt"].cpu() - tgt).abs().max().item() < PIXEL_TOL
            assert (batch["ref"].cpu() - ref).abs().max().item() < PIXEL_TOL
            assert torch.allclose(batch["K"].cpu(), K, atol=1e-5)
            seen.extend(ids)
            n += 1
        assert n == 2
    assert len(set(seen)) == 12


def test_loader_feeds_a_training_step(tmp_path):
    from coivo_amd import data as D, nn as hnn, optim
    ds = D.SequenceFolder(make_tree(str(tmp_path), seqs=(("a", 5, (64, 96)),)))
    ld = D.PairLoader(ds, 2, (64, 96), shuffle=False)
    dn, pn = hnn.DepthNet(), hnn.PoseNet()
    opt = optim.FusedAdam([dn, pn])
    losses = []
    for batch in ld:
        assert batch["frames"].data_ptr() == batch["tgt"].data_ptr() and batch["frames"].shape[0] == 2 * batch["tgt"].shape[0]
        loss = hnn.dcdp_forward(dn, pn, None, None, batch["K"], frames=batch["frames"])[0]      # the stacked buffer: no torch.cat
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert len(losses) == 2 and all(np.isfinite(losses))


def test_mixed_frame_sizes_are_refused(tmp_path):
    from coivo_amd import data as D
    ds = D.SequenceFolder(make_tree(str(tmp_path), seqs=(("a", 3, (48, 64)), ("b", 3, (64, 64)))))
    ld = D.PairLoader(ds, 4, (64, 64), shuffle=False)
    with pytest.raises(ValueError):
        next(iter(ld))


@pytest.mark.parametrize("fmt", ["png", "npy"])
def test_decoder_processes_deliver_the_same_batches(tmp_path, fmt):
    """PairLoader(decoders=N): frames decoded by worker processes into shared, host-registered staging buffers -- the batches must
    equal the in-process loader's bit for bit, over two epochs (ring slots reused), and close() must leave nothing behind."""
    from coivo_amd import data as D
    ds = D.SequenceFolder(make_tree(str(tmp_path), seqs=(("a", 9, (60, 80)), ("b", 7, (60, 80))), fmt=fmt))
    ref = D.PairLoader(ds, 2, (64, 96), shuffle=True, seed=5, workers=2)
    rem = D.PairLoader(ds, 2, (64, 96), shuffle=True, seed=5, workers=2, decoders=3)
    try:
        for ep in range(2):
            ref.set_epoch(ep)
            rem.set_epoch(ep)
            n = 0
            for a, b in zip(ref, rem):
                assert torch.equal(a["tgt"], b["tgt"]) and torch.equal(a["ref"], b["ref"]) and torch.equal(a["K"], b["K"])
                n += 1
            assert n == len(ref) == 7
        names = [shm.name for shm in rem._shm.values()]
        assert names
    finally:
        procs = list(rem._procs)
        rem.close()
    assert all(p.poll() is not None for p in procs)
    from multiprocessing import shared_memory
    for nm in names:
        with pytest.raises(FileNotFoundError):
            shared_memory.SharedMemory(name=nm)


def test_decoder_process_errors_reach_the_caller(tmp_path):
    from coivo_amd import data as D
    root = make_tree(str(tmp_path), seqs=(("a", 4, (48, 64)),))
    ds = D.SequenceFolder(root)
    import os
    victim = sorted(os.listdir(os.path.join(root, "a")))[1]
    with open(os.path.join(root, "a", victim), "wb") as f:
        f.write(b"not an image")
    ld = D.PairLoader(ds, 1, (64, 96), shuffle=False, decoders=2)
    try:
        with pytest.raises(ValueError, match=victim):
            for _ in ld:
                pass
    finally:
        ld.close()
