"""Host-side input / output pipeline around the try-on engine: what keeps the scripts' data loading and image
saving from becoming the serial bottleneck once the denoising loop runs at >= 1.5 images/s per GPU.

  * ShardedSampler     rank r of `world` processes iterates images r, r + world, r + 2*world, ... of the reference's datasets
                       (`VitonHDTestDataset`, /root/reference/inference.py:75-196,303-314, builds one unsharded DataLoader per process
                       and the reference has no distributed sampler at all).  Every image is visited exactly once
                       across ranks, ranks differ by at most one image, and the per-image generator seeds of idm_vton_amd.dist
                       (seed, GLOBAL image index) make the outputs independent of the world size.
  * shard_dataloaders  installs the sampler into every `torch.utils.data.DataLoader(dataset, shuffle=False, ...)` an unmodified
                       script constructs (tests/dropin_launcher.py does it when WORLD_SIZE > 1).
  * AsyncImageWriter   encodes and writes result images (PIL: PNG / JPEG) on worker threads while the GPU already denoises the next
                       batch (inference.py:415-419 encodes each image synchronously between two pipeline calls: ~25 ms per 768x1024 PNG,
                       i.e. 4-8 % of a 2-image call).  Device -> host copies use a pinned staging buffer and a side stream, so
                       `submit()` returns as soon as the copy is queued; `flush()` (also registered atexit) joins everything and
                       re-raises the first error.  Bytes on disk are identical to the synchronous path.
"""
import atexit
import os
import queue
import threading

import torch


class ShardedSampler(torch.utils.data.Sampler):
    """Indices rank, rank + world, ... of a dataset of `n` items (order preserved, no padding, no duplication)."""

    def __init__(self, n, rank, world):
        if not (0 <= rank < world):
            raise ValueError(f"rank {rank} outside world of {world}")
        self.n, self.rank, self.world = int(n), int(rank), int(world)

    def __iter__(self):
        return iter(range(self.rank, self.n, self.world))

    def __len__(self):
        return max(0, (self.n - self.rank + self.world - 1) // self.world)

    def global_index(self, local_position):
        """Position in this rank's iteration order -> dataset index (the index per-image seeds are derived from)."""
        return self.rank + local_position * self.world


def shard_dataloaders(rank, world):
    """Make every map-style `DataLoader(dataset, shuffle=False)` built from now on iterate this rank's shard.  Returns a function
    that undoes the patch.  No-op (returns a no-op) for world == 1."""
    if world <= 1:
        return lambda: None
    DL = torch.utils.data.DataLoader
    orig_init = DL.__init__

    def init(self, dataset, *a, **kw):
        if kw.get("sampler") is None and kw.get("batch_sampler") is None and not kw.get("shuffle", False) and hasattr(dataset, "__len__"):
            kw["sampler"] = ShardedSampler(len(dataset), rank, world)
            kw.pop("shuffle", None)
        orig_init(self, dataset, *a, **kw)

    DL.__init__ = init

    def undo():
        DL.__init__ = orig_init
    return undo


def _to_uint8_hwc(t):
    """CxHxW (or 1xCxHxW) float in [0, 1] -> HxWxC uint8 tensor; the rounding of torchvision.utils.save_image (x*255 + 0.5, clamp).
    torchvision writes a GRID for a batch of N > 1 images; the try-on scripts save one image per call (inference.py:417-419), and a
    silent `t[0]` would drop N - 1 results, so a real batch is an error here."""
    t = t.detach()
    if t.ndim == 4:
        if t.shape[0] != 1:
            raise ValueError(f"save_image: a batch of {t.shape[0]} images (torchvision would write a grid); pass one CxHxW image per call")
        t = t[0]
    if t.ndim != 3:
        raise ValueError(f"save_image: expected CxHxW, got shape {tuple(t.shape)}")
    return t.float().mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(torch.uint8)


class AsyncImageWriter:
    def __init__(self, workers=2, max_pending=16):
        self._q = queue.Queue(maxsize=max_pending)
        self._err = []
        self._threads = [threading.Thread(target=self._run, daemon=True) for _ in range(max(1, workers))]
        self._side = None
        for th in self._threads:
            th.start()
        atexit.register(self.flush)

    def _run(self):
        from PIL import Image
        while True:
            item = self._q.get()
            try:
                if item is None:
                    return
                host, event, path, kw = item
                if event is not None:
                    event.synchronize()                   # the device -> pinned copy of THIS image has landed
                a = host.numpy()
                Image.fromarray(a[:, :, 0] if a.shape[2] == 1 else a).save(path, **kw)
            except Exception as e:                        # surfaced by flush()
                self._err.append(e)
            finally:
                self._q.task_done()

    def submit(self, tensor, path, **save_kwargs):
        """Queue `tensor` (CxHxW float in [0,1], CPU or GPU) to be written to `path`; returns immediately."""
        if isinstance(path, (str, os.PathLike)):
            d = os.path.dirname(os.fspath(path))
            if d:
                os.makedirs(d, exist_ok=True)
        u8 = _to_uint8_hwc(tensor)
        if u8.is_cuda:
            if self._side is None:
                self._side = torch.cuda.Stream(device=u8.device)
            host = torch.empty(u8.shape, dtype=torch.uint8, pin_memory=True)
            self._side.wait_stream(torch.cuda.current_stream(u8.device))
            with torch.cuda.stream(self._side):
                host.copy_(u8, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self._side)
            u8.record_stream(self._side)
            self._q.put((host, ev, path, save_kwargs))
        else:
            self._q.put((u8.contiguous(), None, path, save_kwargs))

    def flush(self):
        self._q.join()
        if self._err:
            e, self._err = self._err[0], []
            raise e

    def close(self):
        self.flush()
        for _ in self._threads:
            self._q.put(None)
        for th in self._threads:
            th.join()
        self._threads = []


_writer = None


def default_writer():
    global _writer
    if _writer is None:
        _writer = AsyncImageWriter()
    return _writer


def save_image_async(tensor, fp, format=None, **kwargs):
    """Drop-in for torchvision.utils.save_image(single image, path): same bytes on disk, written in the background.  torchvision's grid /
    normalisation keywords (nrow, padding, normalize, value_range, scale_each, pad_value) mean something else to PIL.Image.save: they are
    accepted only at the values that leave a single image untouched."""
    neutral = dict(nrow=8, padding=2, normalize=False, value_range=None, scale_each=False, pad_value=0.0)
    for k, v in kwargs.items():
        if k not in neutral:
            raise TypeError(f"save_image: unknown keyword {k!r}")
        if k in ("normalize", "scale_each", "value_range") and v not in (False, None):
            raise NotImplementedError(f"save_image({k}={v!r}) is not supported by the asynchronous writer")
    default_writer().submit(tensor, fp, **({"format": format} if format else {}))


def flush_or_die():
    """Join the default writer; on a write error print it and end the process with a non-zero status (an atexit hook cannot change
    the exit code, and images lost behind a zero status are the worst outcome for a batch job)."""
    if _writer is None:
        return
    try:
        _writer.flush()
    except Exception as e:                                 # noqa: BLE001
        import sys
        import traceback
        traceback.print_exception(type(e), e, e.__traceback__)
        sys.stdout.flush()                                 # os._exit skips the interpreter's own flush of buffered output
        sys.stderr.flush()
        os._exit(1)
