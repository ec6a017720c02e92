"""Frame-pair input pipeline: sequence folders -> sharded batches of (tgt, ref, K) resident in HBM.

Reference: README.md:13 (the dataset is an external download; its on-disk layout is not described, so the layout below is
[ASSUMED]: oracle/SPEC.md §6d).  Host side decodes (PIL / numpy) on a small thread pool into pinned memory; the GPU side is
one H2D copy on a copy stream followed by csrc/frames.hip (resize + de-interleave + /255), double-buffered so the next
batch is decoded and uploaded while the current one trains.  One process per GPU: `rank` / `world_size` shard the pairs
with no communication (a seeded permutation every rank computes identically).
"""
from __future__ import annotations

import os
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib

FRAME_EXT = (".png", ".jpg", ".jpeg", ".bmp", ".npy")


def default_intrinsics(h: int, w: int) -> torch.Tensor:
    """The synthetic generator's pinhole (synth.intrinsics) for sequences without a cam.txt."""
    K = torch.zeros(3, 3)
    K[0, 0] = K[1, 1] = 0.8 * w
    K[0, 2], K[1, 2], K[2, 2] = (w - 1) / 2.0, (h - 1) / 2.0, 1.0
    return K


def resize_intrinsics(K: torch.Tensor, hw_from: Tuple[int, int], hw_to: Tuple[int, int]) -> torch.Tensor:
    """spec: resize_intrinsics -- pixel centres sit on integers, so x -> (x + 1/2) s - 1/2."""
    (h, w), (H, W) = hw_from, hw_to
    sy, sx = H / h, W / w
    K2 = K.clone()
    K2[..., 0, 0] = K[..., 0, 0] * sx
    K2[..., 1, 1] = K[..., 1, 1] * sy
    K2[..., 0, 2] = (K[..., 0, 2] + 0.5) * sx - 0.5
    K2[..., 1, 2] = (K[..., 1, 2] + 0.5) * sy - 0.5
    return K2


def read_frame(path: str) -> np.ndarray:
    """-> [h,w,3] uint8 RGB."""
    if path.endswith(".npy"):
        a = np.load(path)
    else:
        from PIL import Image
        with Image.open(path) as im:
            a = np.asarray(im.convert("RGB"))
    if a.ndim != 3 or a.shape[2] != 3 or a.dtype != np.uint8:
        raise ValueError(f"{path}: expected an 8-bit RGB frame [h,w,3], got {a.dtype} {a.shape}")
    return a


class SequenceFolder:
    """root/<sequence>/<frame>.{png,jpg,bmp,npy} (+ optional cam.txt: 9 numbers, K at native resolution).
    Sample i = (frame k, frame k+skip) of one sequence; frames in lexicographic order."""

    def __init__(self, root: str, skip: int = 1):
        if skip < 1:
            raise ValueError("skip must be >= 1")
        if not os.path.isdir(root):
            raise FileNotFoundError(root)
        self.root, self.skip = root, skip
        self.sequences: List[Tuple[str, List[str], Optional[torch.Tensor]]] = []
        self.pairs: List[Tuple[int, int]] = []
        for name in sorted(os.listdir(root)):
            d = os.path.join(root, name)
            if not os.path.isdir(d):
                continue
            frames = sorted(f for f in os.listdir(d) if f.lower().endswith(FRAME_EXT))
            if len(frames) <= skip:
                continue
            K = None
            cam = os.path.join(d, "cam.txt")
            if os.path.exists(cam):
                vals = np.loadtxt(cam, dtype=np.float64).reshape(-1)
                if vals.size != 9:
                    raise ValueError(f"{cam}: expected 9 numbers (row-major K), got {vals.size}")
                K = torch.from_numpy(vals.reshape(3, 3)).float()
            s = len(self.sequences)
            self.sequences.append((name, [os.path.join(d, f) for f in frames], K))
            self.pairs.extend((s, k) for k in range(len(frames) - skip))
        if not self.pairs:
            raise ValueError(f"{root}: no sequence with more than {skip} frame(s)")

    def __len__(self) -> int:
        return len(self.pairs)

    def __getitem__(self, i: int) -> Dict[str, object]:
        s, k = self.pairs[i]
        name, frames, K = self.sequences[s]
        tgt, ref = read_frame(frames[k]), read_frame(frames[k + self.skip])
        if tgt.shape != ref.shape:
            raise ValueError(f"{name}: frames {k} and {k + self.skip} differ in size ({tgt.shape} vs {ref.shape})")
        if K is None:
            K = default_intrinsics(tgt.shape[0], tgt.shape[1])
        return {"tgt": tgt, "ref": ref, "K": K, "sequence": name, "index": k}


def shard_indices(n: int, batch: int, rank: int, world_size: int, *, shuffle: bool, seed: int, epoch: int) -> List[int]:
    """The pairs rank `rank` trains on in `epoch`: a permutation every rank computes identically, truncated to a multiple
    of world_size*batch (every rank runs the same number of steps: the gradient all-reduce needs all of them), then every
    world_size-th pair starting at `rank`."""
    if not (0 <= rank < world_size) or batch < 1:
        raise ValueError("shard_indices: bad rank / world_size / batch")
    if shuffle:
        g = torch.Generator().manual_seed(seed * 1000003 + epoch)
        order = torch.randperm(n, generator=g).tolist()
    else:
        order = list(range(n))
    usable = (n // (world_size * batch)) * world_size * batch
    return order[rank:usable:world_size]


class PairLoader:
    """Iterating yields dict(tgt, ref [B,3,H,W] fp32 in [0,1] on `device`, K [B,3,3] at (H,W)).  Every batch's frames must
    share one native size (a sequence folder from one camera does)."""

    def __init__(self, dataset: SequenceFolder, batch_size: int, size: Tuple[int, int], *, rank: int = 0,
                 world_size: int = 1, shuffle: bool = True, seed: int = 0, device="cuda", workers: int = 4):
        if size[0] % 32 or size[1] % 32:
            raise ValueError("size (H, W) must be multiples of 32 (DepthNet)")
        self.ds, self.B, self.size = dataset, batch_size, tuple(size)
        self.rank, self.world, self.shuffle, self.seed = rank, world_size, shuffle, seed
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("PairLoader: frames are converted by a HIP kernel; device must be a GPU (no CPU fallback)")
        self.epoch = 0
        self.pool = ThreadPoolExecutor(max_workers=max(1, workers))
        self._stager = ThreadPoolExecutor(max_workers=1)
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self._pinned: Dict[Tuple[int, int, int], List[torch.Tensor]] = {}
        self._flip = 0
        self._uploaded: Dict[int, torch.cuda.Event] = {}

    def set_epoch(self, epoch: int) -> None:
        self.epoch = epoch

    def __len__(self) -> int:
        return len(shard_indices(len(self.ds), self.B, self.rank, self.world, shuffle=False, seed=0, epoch=0)) // self.B

    # ---- host half: decode into pinned memory ------------------------------------------------ #
    def _stage(self, idx: Sequence[int]):
        items = list(self.pool.map(self.ds.__getitem__, idx))
        h, w = items[0]["tgt"].shape[:2]
        for it in items:
            if it["tgt"].shape[:2] != (h, w):
                raise ValueError(f"batch mixes frame sizes: {it['sequence']} is {it['tgt'].shape[:2]}, expected {(h, w)}")
        key = (len(items), h, w)
        bufs = self._pinned.setdefault(key, [])
        while len(bufs) < 2:
            bufs.append(torch.empty(2 * len(items), h, w, 3, dtype=torch.uint8).pin_memory())
        buf = bufs[self._flip]
        self._flip ^= 1
        ev = self._uploaded.pop(buf.data_ptr(), None)
        if ev is not None:
            ev.synchronize()          # the H2D copy that last read this pinned buffer (two batches ago) must be over
        view = buf.numpy()
        for j, it in enumerate(items):
            view[j] = it["tgt"]
            view[len(items) + j] = it["ref"]
        K = torch.stack([resize_intrinsics(it["K"], (h, w), self.size) for it in items])
        return buf, K, (h, w)

    # ---- device half: upload + convert on the copy stream ------------------------------------ #
    def _upload(self, staged):
        buf, K, (h, w) = staged
        lib = _lib.load()
        n = buf.shape[0]
        H, W = self.size
        with torch.cuda.stream(self.copy_stream):
            raw = buf.to(self.device, non_blocking=True)
            out = torch.empty(n, 3, H, W, device=self.device, dtype=torch.float32)
            _lib.check(lib.colvo_frames_u8_to_f32(_lib.ptr(raw), n, h, w, H, W, _lib.ptr(out), _lib.stream_ptr()),
                       "colvo_frames_u8_to_f32")
            Kd = K.to(self.device, non_blocking=True)
            done = torch.cuda.Event()
            done.record(self.copy_stream)
        self._uploaded[buf.data_ptr()] = done
        return out, Kd, done, raw

    def __iter__(self) -> Iterator[Dict[str, torch.Tensor]]:
        idx = shard_indices(len(self.ds), self.B, self.rank, self.world, shuffle=self.shuffle, seed=self.seed,
                            epoch=self.epoch)
        batches = [idx[i:i + self.B] for i in range(0, len(idx), self.B)]
        # decode of batch k+1 (stager thread + its worker pool) overlaps the upload and the training step of batch k
        fut = self._stager.submit(self._stage, batches[0]) if batches else None
        pending = None
        for step in range(len(batches) + 1):
            nxt = None
            if step < len(batches):
                staged = fut.result()
                fut = self._stager.submit(self._stage, batches[step + 1]) if step + 1 < len(batches) else None
                nxt = self._upload(staged)
            if pending is not None:
                out, Kd, done, raw = pending
                torch.cuda.current_stream(self.device).wait_event(done)
                for t in (out, Kd, raw):
                    t.record_stream(torch.cuda.current_stream(self.device))
                B = out.shape[0] // 2
                yield {"tgt": out[:B], "ref": out[B:], "K": Kd}
            pending = nxt
