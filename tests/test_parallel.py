"""The N > 1 path on CPU: two gloo ranks each render their share of the samples and the radiance
sums are reduced onto rank 0.  The renderer injected here is the CPU oracle (the product's
renderer needs a GPU); what is under test is the sharding and reduction code of
raytracer_amd/parallel.py that bench.py runs with RCCL."""
import os
import socket
import sys

import numpy as np
import pytest

from raytracer_amd import parallel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sample_range_partitions():
    for n in (1, 7, 16, 256, 1000):
        for world in (1, 2, 3, 8):
            got = [parallel.sample_range(r, world, n) for r in range(world)]
            assert got[0][0] == 0 and got[-1][1] == n
            assert all(got[i][1] == got[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in got]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        parallel.sample_range(2, 2, 10)


def test_rank_params_weak_and_strong():
    p = parallel.rank_params(64, 48, 4, 4, rank=1, world=2, weak=True)
    assert (p.spp_x, p.spp_y, p.sample_begin, p.sample_end) == (4, 8, 16, 32)
    p = parallel.rank_params(64, 48, 4, 4, rank=1, world=2, weak=False)
    assert (p.spp_x, p.spp_y, p.sample_begin, p.sample_end) == (4, 4, 8, 16)


def test_rank_params_tile_split():
    p = parallel.rank_params(64, 48, 4, 4, rank=1, world=2, split="tile")
    assert (p.tile_step, p.tile_phase, p.sample_begin, p.sample_end) == (2, 1, 0, 16)
    # an image whose width is no multiple of 8 * world: the sample split instead
    p = parallel.rank_params(72, 48, 4, 4, rank=1, world=2, split="tile")
    assert (p.tile_step, p.sample_begin, p.sample_end) == (0, 8, 16)
    assert parallel.tile_share_possible(1920, 1080, 8) and not parallel.tile_share_possible(1920, 1080, 7)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_path, split="sample"):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    from raytracer_amd import scenes
    from tests import orc
    dist.init_process_group("gloo", rank=rank, world_size=world)
    size = 24 if split == "sample" else 32
    sc = scenes.cornell(size, size)
    flat = sc.initKdTree(1).flatten(1)

    def render(p, image):
        img, rs, _, _ = orc.oracle_render(flat.desc, p, 0, threads=1)
        image += torch.from_numpy(img)
        return rs

    params = parallel.rank_params(size, size, 4, 4, rank, world, max_depth=3, seed=11, weak=False, split=split)
    image = torch.zeros((size, size, 3), dtype=torch.float32)
    rs = parallel.render_frame(render, params, image, dist)
    rays = torch.tensor([float(rs.rays_closest + rs.rays_any)], dtype=torch.float64)
    dist.all_reduce(rays)
    if rank == 0:
        np.savez(out_path, image=image.numpy(), rays=rays.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sample_split_equals_single_process(built, tmp_path):
    import torch.multiprocessing as mp

    from raytracer_amd import scenes
    from raytracer_amd.host import render_params
    from tests import orc
    out = str(tmp_path / "rank0.npz")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    z = np.load(out)
    sc = scenes.cornell(24, 24)
    flat = sc.initKdTree(1).flatten(1)
    full, rs, _, _ = orc.oracle_render(flat.desc, render_params(24, 24, 4, 4, max_depth=3, seed=11), 0, threads=2)
    # same sample set; only the order of the float additions differs (two partial sums)
    assert np.allclose(z["image"], full, rtol=1e-5, atol=1e-6)
    assert int(z["rays"][0]) == rs.rays_closest + rs.rays_any


def test_two_rank_tile_split_equals_single_process(built, tmp_path):
    """The interleaved tile shares of two ranks, reduced with the same SUM: the frame of one process bit
    for bit (every pixel is rendered by one rank, the other adds zero), and the same rays."""
    import torch.multiprocessing as mp

    from raytracer_amd import scenes
    from raytracer_amd.host import render_params
    from tests import orc
    out = str(tmp_path / "rank0.npz")
    mp.spawn(_worker, args=(2, _free_port(), out, "tile"), nprocs=2, join=True)
    z = np.load(out)
    sc = scenes.cornell(32, 32)
    flat = sc.initKdTree(1).flatten(1)
    full, rs, _, _ = orc.oracle_render(flat.desc, render_params(32, 32, 4, 4, max_depth=3, seed=11), 0, threads=2)
    assert np.array_equal(z["image"], full) and full.max() > 0
    assert int(z["rays"][0]) == rs.rays_closest + rs.rays_any
