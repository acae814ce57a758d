"""GPU: the N>1 path with the real kernels -- two ranks (gloo rendezvous on 127.0.0.1, both on the one GPU of the test
box) render disjoint pixel ranges of a frame through the C ABI and gather them; the result must be bit-identical to the
single-rank render (rays are independent, every kernel's per-ray arithmetic is position independent)."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_path, n_pixels, H, W):
    import torch.distributed as dist
    from tgtc_style_amd import models, parallel, rendering, synth, utils
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)

    class A:
        use_viewdir, act_type = True, "relu"
        embed_freq_coor, embed_freq_dir = 10, 4
        netdepth = netdepth_fine = 8
        netwidth = netwidth_fine = 256
        precision = "fp16x3"

    nets = []
    for seed, mode in ((0, "coarse"), (1, "fine")):
        m = models.StyleNerf(A, mode=mode)
        m.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.nerf_state(seed).items()})
        nets.append(m.cuda())
    r = rendering.RayRenderer(*nets)
    focal, pose = synth.fern_intrinsics(H, W), synth.spiral_pose(4)

    def make_rays(first, n):
        return utils.gen_rays(H, W, focal, pose, first_pixel=first, n=n)

    def render(o, d):
        out = r.render(o, d, 128, 64)
        return out["rgb"].cpu(), out["t"].cpu()          # gloo gathers host tensors; RCCL would take them on the device

    frame = parallel.render_frame_sharded(render, make_rays, n_pixels, rank, world, dist)
    if rank == 0:
        rgb, t = render(*make_rays(0, n_pixels))
        whole = torch.cat([rgb, t[:, None]], 1)
        np.save(out_path, np.array([int(torch.equal(frame, whole)), frame.shape[0], int(torch.isfinite(frame).all())]))
    dist.barrier()
    dist.destroy_process_group()


# an even split and a ragged one of a 400-wide frame; the WHOLE 504x378 trex frame of BASELINE config 4 (190 512 rays)
@pytest.mark.parametrize("n_pixels,H,W", [(40 * 400, 400, 400), (4001, 400, 400), (504 * 378, 378, 504)])
def test_two_ranks_equal_one(tmp_path, n_pixels, H, W):
    import torch.multiprocessing as mp
    out = str(tmp_path / "result.npy")
    mp.spawn(_worker, args=(2, _free_port(), out, n_pixels, H, W), nprocs=2, join=True)
    same, rows, finite = np.load(out)
    assert rows == n_pixels and finite == 1 and same == 1


def _nccl_worker(_rank, port, out_path):
    """One rank, backend nccl (= RCCL on ROCm); the process group comes up before the first HIP call of the process."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    torch.cuda.set_device(0)
    from tgtc_style_amd import parallel
    ok = []
    # rays sharding: even split -> all_gather_into_tensor straight into the frame
    local = torch.arange(4000 * 4, dtype=torch.float32, device="cuda").reshape(4000, 4)
    rows = parallel.gather_rows(local, 4000, 0, 1, dist, always_collective=True)
    ok.append(rows.is_cuda and torch.equal(rows, local) and rows.data_ptr() != local.data_ptr())
    # frames sharding (bench.py's per-frame gather), synchronously and through the side stream
    frames = parallel.gather_frames(local, 1, dist)
    ok.append(frames.is_cuda and torch.equal(frames, local))
    g = parallel.AsyncGather(lambda x: parallel.gather_frames(x, 1, dist))
    bufs = [torch.empty_like(local) for _ in range(2)]
    last = None
    for i in range(5):          # the compute stream keeps producing while the gathers run beside it
        g.reusable(i)
        bufs[i % 2].copy_(local + float(i))
        g.submit(bufs[i % 2])
        last = local + float(i)
    out = g.result()
    torch.cuda.synchronize()
    ok.append(torch.equal(out, last))
    np.save(out_path, np.array([int(all(ok)), int(dist.get_backend() == "nccl")]))
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_branch_executes_on_device_tensors(tmp_path):
    """The `nccl` branch of parallel.py / bench.py (all_gather_into_tensor on device tensors) in a group of one: the only
    RCCL configuration a one-GPU box can run.  Exercises gather_rows, gather_frames and the side-stream AsyncGather."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "nccl.npy")
    mp.spawn(_nccl_worker, args=(_free_port(), out), nprocs=1, join=True)
    ok, is_nccl = np.load(out)
    assert ok == 1 and is_nccl == 1


def test_bench_two_ranks_gloo_rehearsal():
    """bench.py's N = 2 path (barriers, max-over-ranks timing, side-stream gather) with two gloo ranks on the one GPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TGTC_DIST_BACKEND="gloo")
    for sharding in ("frames", "rays"):
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                            "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                            "--sharding", sharding, "--alt-precision", "", "--configs", "", "--cpu-rays", "0"],
                           capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert line["n_gpus"] == 2 and line["value"] > 0 and line["scaling"] == ("weak" if sharding == "frames" else "strong")
        assert line["config"]["rays_per_step"] == (320000 if sharding == "frames" else 160000)
