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


def gather_rows(local, n_total, rank, world, dist=None):
    """local: [n_local, C] rows of this rank's shard_range -> [n_total, C] on every rank."""
    if world == 1:
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
