"""mrt_intersect_noalloc (the entry point that allocates no device memory) against mrt_intersect.

Both are called through the C-ABI on the same device-resident rays; every output array must be
bit-identical.  Two calls in a row, the second with a smaller batch, for closest-hit and any-hit
semantics, plus the oracle on the first batch so that the pair cannot be wrong together.
"""
import ctypes as C

import numpy as np
import pytest

from raytracer_amd import abi, scenes
from raytracer_amd.host import DeviceScene
from tests import orc

pytestmark = pytest.mark.gpu


def _noalloc(dev, cols, n, any_hit, torch):
    prim = torch.full((n,), -7, dtype=torch.int32, device="cuda:0")
    t = torch.full((n,), -7.0, dtype=torch.float32, device="cuda:0")
    u = torch.full_like(t, -7.0)
    v = torch.full_like(t, -7.0)
    rays = abi.Rays(*[c.data_ptr() for c in cols])
    hits = abi.Hits(prim.data_ptr(), t.data_ptr(), u.data_ptr(), v.data_ptr())
    stream = torch.cuda.current_stream().cuda_stream
    st = abi.hip().mrt_intersect_noalloc(dev._h, n, C.byref(rays), C.byref(hits), abi.MRT_ANY_HIT if any_hit else abi.MRT_CLOSEST_HIT, stream)
    assert st == abi.MRT_OK, abi.hip().mrt_last_error().decode()
    return prim, t, u, v


@pytest.mark.parametrize("any_hit", [False, True])
def test_noalloc_matches_intersect(built, any_hit):
    import torch
    sc = scenes.small_bunny(96, 96)
    tree = sc.initKdTree()
    flat = tree.flatten(1)
    dev = DeviceScene(flat)
    o1, d1 = orc.camera_rays(flat.desc, 96, 96, seed=11)
    o2, d2 = orc.random_rays(flat.desc, 30000, seed=12)
    o, d = np.concatenate([o1, o2]), np.concatenate([d1, d2])
    # first call: the whole batch; second call: a smaller one (a tail that is not a multiple of a chunk)
    for n in (len(o), 5003):
        oo, dd = o[:n], d[:n]
        tmin = np.zeros(n, np.float32)
        tmax = np.full(n, 1e5, np.float32)
        to, td = torch.from_numpy(oo).cuda(), torch.from_numpy(dd).cuda()
        ttmin, ttmax = torch.from_numpy(tmin).cuda(), torch.from_numpy(tmax).cuda()
        cols = [to[:, 0].contiguous(), to[:, 1].contiguous(), to[:, 2].contiguous(), td[:, 0].contiguous(),
                td[:, 1].contiguous(), td[:, 2].contiguous(), ttmin, ttmax]
        got = _noalloc(dev, cols, n, any_hit, torch)
        want = dev.intersect(to, td, ttmin, ttmax, any_hit=any_hit)
        torch.cuda.synchronize()
        for g, w, name in zip(got, want, ("prim", "t", "u", "v")):
            g, w = g.cpu().numpy(), w.cpu().numpy()
            assert np.array_equal(g.view(np.uint32), w.view(np.uint32)), "%s differs on %d of %d rays" % (name, int(np.sum(g.view(np.uint32) != w.view(np.uint32))), n)
        ref = orc.oracle_intersect(flat.desc, oo, dd, tmin, tmax, any_hit)
        prim = got[0].cpu().numpy()
        assert np.array_equal(prim >= 0, ref["prim"] >= 0)
        if not any_hit:
            assert np.array_equal(prim, ref["prim"])
        assert int((prim >= 0).sum()) > 100
    dev.close()


def test_noalloc_rejects_oversize(built):
    sc = scenes.cornell(32, 32)
    flat = sc.initKdTree().flatten(1)
    dev = DeviceScene(flat)
    rays = abi.Rays(*[8] * 8)  # never dereferenced: the size check comes first
    hits = abi.Hits(8, 8, 8, 8)
    st = abi.hip().mrt_intersect_noalloc(dev._h, 1 << 31, C.byref(rays), C.byref(hits), abi.MRT_CLOSEST_HIT, None)
    assert st == 4  # MRT_ERR_UNSUPPORTED
    assert b"MRT_NOALLOC_MAX_RAYS" in abi.hip().mrt_last_error()
    dev.close()


def test_noalloc_allocates_nothing_and_can_be_captured(built):
    """Free device memory is the same before and after a call (mrt_intersect with device buffers
    allocates nothing either once the scene exists, so the comparison is with the state right before),
    and the call can be recorded into a HIP graph -- which refuses hipMalloc -- and replayed: the
    replayed launch writes the same hits."""
    import torch
    sc = scenes.small_bunny(64, 64)
    flat = sc.initKdTree().flatten(1)
    dev = DeviceScene(flat)
    o, d = orc.camera_rays(flat.desc, 64, 64, seed=3)
    n = len(o)
    to, td = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()
    cols = [to[:, 0].contiguous(), to[:, 1].contiguous(), to[:, 2].contiguous(), td[:, 0].contiguous(), td[:, 1].contiguous(),
            td[:, 2].contiguous(), torch.zeros(n, device="cuda"), torch.full((n,), 1e5, device="cuda")]
    want = _noalloc(dev, cols, n, False, torch)  # warm: the kernel's code object is loaded, the occupancy asked for
    torch.cuda.synchronize()
    out = [torch.full((n,), -7, dtype=torch.int32, device="cuda:0")] + [torch.full((n,), -7.0, device="cuda:0") for _ in range(3)]
    rays = abi.Rays(*[c.data_ptr() for c in cols])
    hits = abi.Hits(*[x.data_ptr() for x in out])
    free_before = torch.cuda.mem_get_info()[0]
    st = abi.hip().mrt_intersect_noalloc(dev._h, n, C.byref(rays), C.byref(hits), abi.MRT_CLOSEST_HIT, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert st == abi.MRT_OK and torch.cuda.mem_get_info()[0] == free_before
    for g, w in zip(out, want):
        assert torch.equal(g, w)
    for x in out:
        x.fill_(-7)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        st = abi.hip().mrt_intersect_noalloc(dev._h, n, C.byref(rays), C.byref(hits), abi.MRT_CLOSEST_HIT, torch.cuda.current_stream().cuda_stream)
    assert st == abi.MRT_OK, abi.hip().mrt_last_error().decode()
    torch.cuda.synchronize()
    assert int((out[0] == -7).all())  # capture records, it does not run
    graph.replay()
    torch.cuda.synchronize()
    for g, w in zip(out, want):
        assert torch.equal(g, w)
    dev.close()
