"""CPU (gloo, world_size 2 and 3) tests of the multi-GPU sharding logic: a frame rendered in shards and
re-assembled by all-gather is bit-identical to the frame rendered in one piece, including ragged splits."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from tgtc_style_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_rays(first, n):
    idx = torch.arange(first, first + n, dtype=torch.float64)
    return torch.stack([idx, idx * 0.5, -torch.ones_like(idx)], 1), torch.stack([idx * 1e-3, idx * 2e-3, 2 + 0 * idx], 1)


def _fake_render(o, d):   # any per-ray function: rays are independent, like the real path
    rgb = torch.sigmoid(torch.stack([o[:, 0] * 1e-2, d[:, 1], o[:, 1] * d[:, 0]], 1)).float()
    return rgb, (o[:, 0] * 1e-4 + d[:, 0]).float()


def _worker(rank, world, port, n_pixels, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    frame = parallel.render_frame_sharded(_fake_render, _fake_rays, n_pixels, rank, world, dist)
    q.put((rank, frame.numpy()))     # by value: tensors would travel as shared-memory handles
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_pixels", [(2, 400), (2, 401), (3, 1000)])
def test_sharded_frame_equals_whole(world, n_pixels):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_pixels, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rgb, t = _fake_render(*_fake_rays(0, n_pixels))
    whole = torch.cat([rgb, t[:, None]], 1)
    for r in range(world):
        assert torch.equal(torch.from_numpy(got[r]), whole)


def test_shard_ranges_partition():
    for n in (0, 1, 7, 160000, 190512):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert parallel.shard_range(190512, 3, 8) == (3 * 23814, 4 * 23814)      # trex 504x378 splits exactly
    assert sorted(sum((parallel.frames_of_rank(120, r, 8) for r in range(8)), [])) == list(range(120))
