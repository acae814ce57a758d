"""Asynchronous image files for the render drivers ("only 8-bit images cross PCIe; async
PNG/JPEG encode on host threads").

The reference writes every finished frame synchronously (rendering.py:209-218, :362-364 `imageio.imwrite`;
trans_test.py:166-174 `save_image`): device -> host copy, PNG encode and disk write sit between two frames' kernels.
Here the 8-bit image produced by the device epilogue is copied into pinned host memory with a non-blocking copy, an
event is recorded behind the copy, and a small thread pool waits for the event, encodes (PIL releases the GIL while it
compresses) and writes, while the calling thread already enqueues the next frame.  `drain()` joins the pool and
re-raises the first failure; every driver calls it before it returns, so a returned driver means files on disk.

Host plumbing only: no arithmetic on pixel values happens here.
"""
import os
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch


def _encode(path, arr):
    """Encode beside the target and rename into place: a run killed mid-encode leaves `<name>.tmp`, never a truncated
    image that the resume-by-skipping of the drivers (rendering.py:267-270) would take for a finished frame."""
    from PIL import Image
    tmp = path + ".tmp"
    Image.fromarray(arr).save(tmp, format=os.path.splitext(path)[1][1:].upper().replace("JPG", "JPEG") or None)
    os.replace(tmp, path)


class ImageWriter:
    def __init__(self, workers=4):
        self._pool = ThreadPoolExecutor(max_workers=workers, thread_name_prefix="tgtc-image")
        self._futures = []
        self._lock = threading.Lock()

    def save(self, path, image):
        """image: uint8 array-like (numpy, or a CUDA / CPU torch tensor) already shaped [h,w] or [h,w,3]."""
        if isinstance(image, torch.Tensor) and image.is_cuda:
            assert image.dtype == torch.uint8
            host = torch.empty(image.shape, dtype=torch.uint8, pin_memory=True)
            host.copy_(image, non_blocking=True)
            done = torch.cuda.Event()
            done.record()

            def job(keep=image):        # `keep` holds the device tensor until the copy has completed
                done.synchronize()
                _encode(path, host.numpy())
        else:
            arr = np.ascontiguousarray(image.numpy() if isinstance(image, torch.Tensor) else np.asarray(image), dtype=np.uint8)

            def job():
                _encode(path, arr)
        with self._lock:
            self._futures.append(self._pool.submit(job))
        if os.environ.get("TGTC_SYNC_IMAGES") == "1":      # debugging: the reference's synchronous behaviour
            self.drain()

    def drain(self):
        with self._lock:
            futures, self._futures = self._futures, []
        err = None
        for f in futures:
            try:
                f.result()
            except Exception as e:          # keep joining: a failed file must not leave others half written
                err = err or e
        if err is not None:
            raise err


_writer = None


def writer():
    """The process-wide writer (threads are created on first use)."""
    global _writer
    if _writer is None:
        _writer = ImageWriter()
    return _writer
