"""Frame-pair input pipeline: sequence folders -> sharded batches of (tgt, ref, K) resident in HBM.

Reference: README.md:13 (the dataset is an external download; its on-disk layout is not described, so the layout below is
[ASSUMED]: oracle/SPEC.md §6d).  Host side decodes (PIL / numpy) on a small thread pool into pinned memory; the GPU side is
one H2D copy on a copy stream followed by csrc/frames.hip (resize + de-interleave + /255); `prefetch` batches are being
decoded while the current one trains.  Raw `.npy` frames take the library's native reader (colvo_read_npy_u8_frames: parallel
pread() straight into the pinned buffer, no interpreter lock): 15-30 k pairs/s at 320x256 on a 16-core box against ~4.5 k through
the interpreter; PNG / JPEG frames are decoded by PIL on the thread pool (~1.0 k pairs/s beside a training loop) or, with
`decoders=N`, by N worker processes writing into shared pinned staging buffers (~3 k pairs/s beside a training loop, N = 10).  One process per GPU: `rank` / `world_size` shard the pairs
with no communication (a seeded permutation every rank computes identically).
"""
from __future__ import annotations

import collections
import os
import queue
import subprocess
import sys
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


def _npy_header(path: str):
    """(data offset, shape, dtype, fortran_order) of a .npy file."""
    with open(path, "rb") as f:
        major, _ = np.lib.format.read_magic(f)
        shape, fortran, dtype = (np.lib.format.read_array_header_1_0(f) if major == 1 else np.lib.format.read_array_header_2_0(f))
        return f.tell(), shape, dtype, fortran


def read_frame_into(path: str, out: np.ndarray) -> None:
    """Decode / read one frame straight into `out` ([h,w,3] uint8, C-contiguous -- a slice of the pinned staging buffer).  Raw
    .npy frames are read with readinto(): no intermediate array, and the interpreter lock is released for the whole read."""
    if path.endswith(".npy"):
        off, shape, dtype, fortran = _npy_header(path)
        if tuple(shape) != out.shape or dtype != np.uint8 or fortran:
            raise ValueError(f"{path}: expected an 8-bit RGB frame {out.shape}, got {dtype} {tuple(shape)}")
        with open(path, "rb", buffering=0) as f:
            f.seek(off)
            mv = memoryview(out).cast("B")
            got = 0
            while got < len(mv):
                n = f.readinto(mv[got:])
                if not n:
                    raise ValueError(f"{path}: truncated file")
                got += n
        return
    a = read_frame(path)
    if a.shape != out.shape:
        raise ValueError(f"{path}: frame is {a.shape}, expected {out.shape}")
    out[...] = a


def read_npy_frames(paths: Sequence[str], out: np.ndarray, threads: int = 8) -> None:
    """Raw frames (`.npy` files of [h,w,3] uint8 arrays) -> out[len(paths), h, w, 3] (uint8, C-contiguous) through the library's
    native reader (colvo_read_npy_u8_frames: parallel pread() into `out`, every header checked against out's frame shape)."""
    import ctypes
    if out.dtype != np.uint8 or out.ndim != 4 or out.shape[0] != len(paths) or out.shape[3] != 3 or not out.flags["C_CONTIGUOUS"]:
        raise ValueError("read_npy_frames: out must be a C-contiguous uint8 array [len(paths), h, w, 3]")
    lib = _lib.load()
    arr = (ctypes.c_char_p * len(paths))(*[os.fsencode(p) for p in paths])
    _lib.check(lib.colvo_read_npy_u8_frames(arr, len(paths), out.shape[1], out.shape[2], out.ctypes.data, int(threads)),
               "colvo_read_npy_u8_frames")


def frame_size(path: str) -> Tuple[int, int]:
    """(h, w) of a frame without decoding it."""
    if path.endswith(".npy"):
        shape = _npy_header(path)[1]
        return int(shape[0]), int(shape[1])
    from PIL import Image
    with Image.open(path) as im:
        return im.height, im.width


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
        self._sizes: Dict[int, Tuple[int, int]] = {}
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

    def pair_info(self, i: int):
        """(tgt path, ref path, K at native size, (h, w), sequence name) of sample i, without reading the frames (the size of a
        sequence's frames is looked up once)."""
        s, k = self.pairs[i]
        name, frames, K = self.sequences[s]
        hw = self._sizes.get(s)
        if hw is None:
            hw = self._sizes[s] = frame_size(frames[0])
        if K is None:
            K = default_intrinsics(hw[0], hw[1])
        return frames[k], frames[k + self.skip], K, hw, name


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
    """Iterating yields dict(tgt, ref [B,3,H,W] fp32 in [0,1] on `device`, K [B,3,3] at (H,W), frames [2B,3,H,W] = the buffer tgt
    and ref are the two halves of).  Every batch's frames must share one native size (a sequence folder from one camera does)."""

    def __init__(self, dataset: SequenceFolder, batch_size: int, size: Tuple[int, int], *, rank: int = 0,
                 world_size: int = 1, shuffle: bool = True, seed: int = 0, device="cuda", workers: int = 8, prefetch: int = 2,
                 own_copy_stream: Optional[bool] = None, decoders: int = 0):
        if size[0] % 32 or size[1] % 32:
            raise ValueError("size (H, W) must be multiples of 32 (DepthNet)")
        self.ds, self.B, self.size = dataset, batch_size, tuple(size)
        self.rank, self.world, self.shuffle, self.seed = rank, world_size, shuffle, seed
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("PairLoader: frames are converted by a HIP kernel; device must be a GPU (no CPU fallback)")
        self.epoch = 0
        # `prefetch` batches are decoded at the same time (one staging thread each, sharing the pool of `workers` decoders):
        # with a single batch in flight the rate is one batch per decode latency, whatever the number of cores
        self.prefetch = max(1, int(prefetch))
        self.workers = max(1, int(workers))
        self.pool = ThreadPoolExecutor(max_workers=max(self.workers, int(decoders)))
        self._stager = ThreadPoolExecutor(max_workers=self.prefetch)
        # The upload (H2D copy + conversion kernel) runs on a copy stream of the loader's own, behind the previous step -- but that
        # is one more active hardware queue, and beside the training step's three (main, the weight-gradient side stream, the
        # library's auxiliary one) a fourth makes the runtime serialise the whole backward pass (measured with this loader: 4.8 ms
        # per step instead of 1.5).  So, by default (own_copy_stream=None):
        #   * single process: own copy stream, and the library's auxiliary stream is switched off (colvo_set_aux_side_streams(0),
        #     as ddp.GradBuckets does for RCCL's stream): 1.61 ms per step with the loader against 1.52 ms on resident tensors;
        #   * under an initialised process group (data parallel: RCCL's stream is the third queue): the upload is enqueued on the
        #     consumer's stream instead (1.80 ms per step: the copy then sits in front of the step).
        if own_copy_stream is None:
            import torch.distributed as dist
            own_copy_stream = not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)
        self.copy_stream = None
        # (With GPU_MAX_HW_QUEUES=2 in the environment before HIP initialises, the runtime folds all streams onto two hardware
        # queues: four streams are then harmless -- measured 5210 pairs/s with the auxiliary stream left on, the resident-tensor
        # rate -- so the auxiliary stream stays.)
        self._queue_claim = None
        if own_copy_stream:
            self.copy_stream = torch.cuda.Stream(device=self.device)
            from . import streams
            self._queue_claim = streams.claim_external_queue("loader copy stream")      # given back by close()
        self._slot_seq = 0                  # ring-slot counter, monotonic across iterations (see __iter__)
        self._inflight: "collections.deque" = collections.deque()
        self._pinned: Dict[Tuple[int, int, int, int], torch.Tensor] = {}      # (slot, 2B, h, w) -> pinned staging buffer
        self._uploaded: Dict[int, torch.cuda.Event] = {}
        # decoders > 0: PNG / JPEG frames are decoded by that many worker PROCESSES (coivo_amd/_decode_worker.py) writing into
        # shared-memory staging buffers registered as pinned memory -- the interpreter lock caps in-process decoding at ~1.5 k
        # pairs/s.  Raw .npy frames never need them (native reader).
        self._shm: Dict[Tuple[int, int, int, int], object] = {}
        self._idle: "queue.Queue" = queue.Queue()
        self._procs: List[subprocess.Popen] = []
        for _ in range(max(0, int(decoders))):
            p = subprocess.Popen([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "_decode_worker.py")],
                                 stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, bufsize=1)
            self._procs.append(p)
            self._idle.put(p)

    def close(self) -> None:
        """Stop the decoder processes, release the shared staging buffers and give the copy stream's hardware queue back to
        the stream policy (also called when the loader is collected)."""
        self._drain()
        if getattr(self, "_queue_claim", None) is not None:
            self._queue_claim.release()
            self._queue_claim = None
        for p in self._procs:
            try:
                p.stdin.close()
                p.wait(timeout=5)
            except Exception:           # noqa: BLE001
                p.kill()
        self._procs = []
        for key, shm in list(self._shm.items()):
            buf = self._pinned.pop(key, None)
            if buf is not None:
                torch.cuda.cudart().cudaHostUnregister(buf.data_ptr())
            del buf
            for fin in (shm.close, shm.unlink):       # unlink even when a view of the block is still alive somewhere
                try:
                    fin()
                except Exception:       # noqa: BLE001
                    pass
        self._shm = {}

    def __del__(self):
        try:
            self.close()
        except Exception:               # noqa: BLE001
            pass

    def _staging_buffer(self, key) -> torch.Tensor:
        """The pinned [2n,h,w,3] uint8 buffer of ring slot key[0]; with decoder processes it lives in shared memory."""
        buf = self._pinned.get(key)
        if buf is not None:
            return buf
        _, n, h, w = key
        if not self._procs:
            buf = torch.empty(2 * n, h, w, 3, dtype=torch.uint8).pin_memory()
        else:
            from multiprocessing import shared_memory
            nbytes = 2 * n * h * w * 3
            shm = shared_memory.SharedMemory(create=True, size=nbytes)
            buf = torch.frombuffer(shm.buf, dtype=torch.uint8, count=nbytes).view(2 * n, h, w, 3)
            rc = torch.cuda.cudart().cudaHostRegister(buf.data_ptr(), nbytes, 0)
            if int(rc) != 0:
                shm.close(); shm.unlink()
                raise RuntimeError(f"PairLoader: cudaHostRegister of the shared staging buffer failed ({rc})")
            self._shm[key] = shm
        self._pinned[key] = buf
        return buf

    def _decode_remote(self, shm_name: str, offset: int, h: int, w: int, path: str) -> None:
        p = self._idle.get()
        try:
            p.stdin.write(f"{shm_name}\t{offset}\t{h}\t{w}\t{path}\n")
            p.stdin.flush()
            reply = p.stdout.readline().rstrip("\n")
        finally:
            self._idle.put(p)
        if reply != "ok":
            raise ValueError(f"{path}: {reply[4:] if reply.startswith('err ') else 'decoder process died'}")

    def set_epoch(self, epoch: int) -> None:
        self.epoch = epoch

    def __len__(self) -> int:
        return len(shard_indices(len(self.ds), self.B, self.rank, self.world, shuffle=False, seed=0, epoch=0)) // self.B

    # ---- host half: decode into pinned memory ------------------------------------------------ #
    def _stage(self, idx: Sequence[int], slot: int):
        """Decode one batch into the pinned buffer of ring slot `slot` (slots are handed out by the iterating thread: a slot is
        staged by one thread at a time and comes round again only after prefetch + 2 batches)."""
        infos = [self.ds.pair_info(i) for i in idx]
        h, w = infos[0][3]
        for info in infos:
            if info[3] != (h, w):
                raise ValueError(f"batch mixes frame sizes: {info[4]} is {info[3]}, expected {(h, w)}")
        n = len(infos)
        key = (slot, n, h, w)
        buf = self._staging_buffer(key)
        ev = self._uploaded.pop(buf.data_ptr(), None)
        if ev is not None:
            ev.synchronize()          # the H2D copy that last read this pinned buffer (a ring turn ago) must be over
        view = buf.numpy()
        # every frame is decoded / read straight into its slot of the pinned buffer (2n independent jobs for the decoder pool)
        paths = [info[0] for info in infos] + [info[1] for info in infos]
        if all(p.endswith(".npy") for p in paths):
            read_npy_frames(paths, view, self.workers)      # native: headers checked and payloads read in C, no interpreter lock
            K = resize_intrinsics(torch.stack([info[2] for info in infos]), (h, w), self.size)
            return buf, K, (h, w)
        if self._procs:                   # one frame per request to whichever decoder process is idle
            shm_name, fb = self._shm[key].name, h * w * 3
            todo = [(j, p) for j, p in enumerate(paths) if not p.endswith(".npy")]
            for j, p in enumerate(paths):
                if p.endswith(".npy"):
                    read_frame_into(p, view[j])
            list(self.pool.map(lambda jp: self._decode_remote(shm_name, jp[0] * fb, h, w, jp[1]), todo))
            K = resize_intrinsics(torch.stack([info[2] for info in infos]), (h, w), self.size)
            return buf, K, (h, w)
        jobs = [(p, view[j]) for j, p in enumerate(paths)]
        nw = min(len(jobs), self.workers)

        def run(chunk):                  # a few frames per pool task: the hand-over to a pool thread costs about as much as a raw read
            for path, out in chunk:
                read_frame_into(path, out)
        list(self.pool.map(run, [jobs[c::nw] for c in range(nw)]))
        K = resize_intrinsics(torch.stack([info[2] for info in infos]), (h, w), self.size)      # one batched call
        return buf, K, (h, w)

    # ---- device half: upload + convert on the copy stream ------------------------------------ #
    def _upload(self, staged):
        buf, K, (h, w) = staged
        lib = _lib.load()
        n = buf.shape[0]
        H, W = self.size
        stream = self.copy_stream if self.copy_stream is not None else torch.cuda.current_stream(self.device)
        with torch.cuda.stream(stream):
            raw = buf.to(self.device, non_blocking=True)
            out = torch.empty(n, 3, H, W, device=self.device, dtype=torch.float32)
            _lib.check(lib.colvo_frames_u8_to_f32(_lib.ptr(raw), n, h, w, H, W, _lib.ptr(out), _lib.stream_ptr()),
                       "colvo_frames_u8_to_f32")
            Kd = K.to(self.device, non_blocking=True)
            done = torch.cuda.Event()
            done.record(stream)
        self._uploaded[buf.data_ptr()] = done
        return out, Kd, done, raw

    def __iter__(self) -> Iterator[Dict[str, torch.Tensor]]:
        idx = shard_indices(len(self.ds), self.B, self.rank, self.world, shuffle=self.shuffle, seed=self.seed,
                            epoch=self.epoch)
        batches = [idx[i:i + self.B] for i in range(0, len(idx), self.B)]
        # decode of batches k+1 .. k+prefetch (staging threads + the decoder pool) overlaps the upload and the training step
        # of batch k; ring of prefetch + 2 pinned buffers: `prefetch` being filled, one being uploaded, one of margin
        ring = self.prefetch + 2
        # An iterator abandoned mid-epoch (break) may have left staging tasks running: wait for them before handing out slots
        # again, and number the slots with a counter that never restarts, so that a new iteration cannot stage into a pinned
        # buffer an old task -- or an upload still in flight -- is using.
        self._drain()
        futs = self._inflight
        submitted = 0

        def submit():
            nonlocal submitted
            if submitted < len(batches):
                futs.append(self._stager.submit(self._stage, batches[submitted], self._slot_seq % ring))
                self._slot_seq += 1
                submitted += 1
        try:
            for _ in range(self.prefetch):
                submit()
            pending = None
            for step in range(len(batches) + 1):
                nxt = None
                if step < len(batches):
                    staged = futs.popleft().result()
                    submit()
                    nxt = self._upload(staged)
                if pending is not None:
                    out, Kd, done, raw = pending
                    torch.cuda.current_stream(self.device).wait_event(done)
                    for t in (out, Kd, raw):
                        t.record_stream(torch.cuda.current_stream(self.device))
                    B = out.shape[0] // 2
                    # "frames": the stacked buffer itself, [target frames | reference frames] = what DepthNet.forward_pair* and
                    # nn.dcdp_forward(frames=...) take -- no torch.cat in the train loop
                    yield {"tgt": out[:B], "ref": out[B:], "K": Kd, "frames": out}
                pending = nxt
        finally:
            self._drain()           # generator closed early (break / exception): no staging task outlives its iterator

    def _drain(self) -> None:
        """Wait for every outstanding staging task (their results are dropped; errors of abandoned tasks are not raised)."""
        q = getattr(self, "_inflight", None)
        while q:
            f = q.popleft()
            if not f.cancel():
                try:
                    f.result()
                except Exception:           # noqa: BLE001
                    pass
