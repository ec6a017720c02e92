"""Image-decoding worker process of data.PairLoader (started as a plain script: it must not import torch or the package).

The interpreter lock caps in-process PNG / JPEG decoding at ~1.5 k frame pairs/s however many threads there are; PairLoader
(decoders=N) starts N of these and hands each frame to an idle one.  Protocol, one request per line on stdin:
    <shared-memory name> TAB <byte offset> TAB <h> TAB <w> TAB <path>
The frame is decoded to 8-bit RGB [h,w,3] straight into the shared-memory block (PairLoader's pinned staging ring); the reply on
stdout is "ok" or "err <message>".  The worker exits when stdin closes.
"""
import sys
from multiprocessing import shared_memory, resource_tracker

import numpy as np


def main():
    from PIL import Image
    blocks = {}
    out = sys.stdout
    for line in sys.stdin:
        try:
            name, off, h, w, path = line.rstrip("\n").split("\t", 4)
            off, h, w = int(off), int(h), int(w)
            shm = blocks.get(name)
            if shm is None:
                shm = blocks[name] = shared_memory.SharedMemory(name=name)
                # the parent owns the block: keep this process's resource tracker from unlinking it at exit
                try:
                    resource_tracker.unregister(shm._name, "shared_memory")
                except Exception:       # noqa: BLE001
                    pass
            dst = np.ndarray((h, w, 3), dtype=np.uint8, buffer=shm.buf, offset=off)
            with Image.open(path) as im:
                a = np.asarray(im.convert("RGB"))
            if a.shape != (h, w, 3):
                raise ValueError(f"frame is {a.shape}, expected {(h, w, 3)}")
            dst[...] = a
            out.write("ok\n")
        except Exception as e:          # noqa: BLE001 -- reported to the parent, which raises
            out.write("err " + f"{type(e).__name__}: {e}".replace("\n", " ") + "\n")
        out.flush()
    for shm in blocks.values():
        shm.close()


if __name__ == "__main__":
    main()
