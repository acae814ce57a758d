"""Ray / frame sharding across the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" is
RCCL on ROCm, "gloo" in the CPU tests).

The render path is embarrassingly parallel: every ray is independent, so ranks take
contiguous pixel ranges of a frame (or whole frames), generate their own rays on the device, render, and one
all-gather of the [rays, 4] RGB+depth rows reassembles the image on every rank.  No other collective exists on
the path.  With equal shards the gather is `all_gather_into_tensor`; ragged tails fall back to padded shards.
"""
import torch


def shard_range(n, rank, world):
    """Contiguous balanced split of range(n): the first n % world ranks get one extra element."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def frames_of_rank(n_frames, rank, world):
    """Round-robin frame ownership (config 5: no per-frame collective)."""
    return list(range(rank, n_frames, world))


def gather_rows(local, n_total, rank, world, dist=None, always_collective=False):
    """local: [n_local, C] rows of this rank's shard_range -> [n_total, C] on every rank.
    always_collective: run the all-gather even in a group of one (tests exercise the RCCL call on a one-GPU box)."""
    if world == 1 and not always_collective:
        return local
    if dist is None:
        import torch.distributed as dist
    base, extra = divmod(n_total, world)
    width = base + (1 if extra else 0)
    if extra == 0:
        out = torch.empty(n_total, local.shape[1], dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous())
        return out
    pad = torch.zeros(width, local.shape[1], dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    buf = torch.empty(world * width, local.shape[1], dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, pad)
    pieces = []
    for r in range(world):
        lo, hi = shard_range(n_total, r, world)
        pieces.append(buf[r * width: r * width + (hi - lo)])
    return torch.cat(pieces, 0)


def gather_frames(local, world, dist):
    """Frames sharding: every rank holds a whole [n, C] image of its own pose -> [world * n, C] on every rank, one
    `all_gather_into_tensor` (RCCL `ncclAllGather` over xGMI under the nccl backend; gloo has no tensor form)."""
    out = torch.empty(world * local.shape[0], local.shape[1], device=local.device, dtype=local.dtype)
    if dist.get_backend() == "nccl":
        dist.all_gather_into_tensor(out, local.contiguous())
    else:
        dist.all_gather(list(out.chunk(world)), local.contiguous())
    return out


class AsyncGather:
    """The per-frame gather on a SIDE stream (the gather is latency, not bandwidth -- 2.5 MB per
    rank -- so it is overlapped with the next frame's render instead of serialised behind it on the compute stream).

        g = AsyncGather(fn)            # fn(local) -> gathered tensor, e.g. lambda x: gather_frames(x, world, dist)
        g.submit(image)                # compute stream: an event behind everything enqueued so far; side stream: wait + fn
        ...enqueue the next frame on the compute stream...
        frame = g.result()             # the compute stream waits for the newest gather (event), returns its tensor

    `image` must not be overwritten while its gather is in flight: keep two image buffers and call `reusable(i)` --
    which makes the compute stream wait for the gather submitted two frames ago -- before writing into buffer i % 2."""

    def __init__(self, fn):
        self.fn = fn
        self.side = torch.cuda.Stream()
        self.done = [None, None]
        self.out = None
        self.n = 0

    def reusable(self, i):
        ev = self.done[i % 2]
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def submit(self, local):
        ready = torch.cuda.Event()
        ready.record()
        with torch.cuda.stream(self.side):
            self.side.wait_event(ready)
            self.out = self.fn(local)
            local.record_stream(self.side)
            done = torch.cuda.Event()
            done.record()
        self.done[self.n % 2] = done
        self.n += 1

    def result(self):
        if self.n:
            torch.cuda.current_stream().wait_event(self.done[(self.n - 1) % 2])
            self.out.record_stream(torch.cuda.current_stream())
        return self.out


def render_frame_sharded(render_rays, make_rays, n_pixels, rank, world, dist=None):
    """Render one frame across `world` ranks.

    make_rays(first_pixel, n) -> (rays_o, rays_d) for that pixel range (generated on this rank's device);
    render_rays(rays_o, rays_d) -> (rgb [n,3], depth [n]).  Returns the whole frame [n_pixels, 4] on every rank.
    """
    lo, hi = shard_range(n_pixels, rank, world)
    o, d = make_rays(lo, hi - lo)
    rgb, t = render_rays(o, d)
    local = torch.cat([rgb, t[:, None]], 1)
    return gather_rows(local, n_pixels, rank, world, dist)
