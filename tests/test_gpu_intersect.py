"""GPU parity of mrt_intersect (through the C-ABI) against the CPU oracle's intersectKdTree.

Bar: primitive ids bit-exact; t and barycentrics are compared bit-exact as well (the kernels
evaluate the same unfused fp32 expression tree), which is stricter than the 1e-5 relative
tolerance the acceptance criterion allows.
"""
import numpy as np
import pytest

from raytracer_amd import scenes
from raytracer_amd.host import DeviceScene
from tests import orc

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _compare(flat, dev, o, d, tmin, tmax, any_hit):
    want = orc.oracle_intersect(flat.desc, o, d, tmin, tmax, any_hit)
    prim, t, u, v = dev.intersect(o, d, tmin, tmax, any_hit=any_hit)
    hit = want["prim"] >= 0
    assert np.array_equal(prim >= 0, hit), "hit/miss differs on %d rays" % int(np.sum((prim >= 0) != hit))
    if any_hit:
        # which occluder is reported is fixed too (first non-emissive in traversal order)
        assert np.array_equal(prim[hit], want["prim"][hit])
        assert np.array_equal(_bits(t[hit]), _bits(want["t"][hit]))
        return int(hit.sum())
    assert np.array_equal(prim, want["prim"]), "%d primitive ids differ" % int(np.sum(prim != want["prim"]))
    assert np.array_equal(_bits(t[hit]), _bits(want["t"][hit]))
    return int(hit.sum())


def _rays(flat, w, h, n_random, seed):
    o1, d1 = orc.camera_rays(flat.desc, w, h, seed=seed)
    o2, d2 = orc.random_rays(flat.desc, n_random, seed=seed + 1)
    o, d = np.concatenate([o1, o2]), np.concatenate([d1, d2])
    n = len(o)
    rng = np.random.default_rng(seed + 2)
    tmin = np.zeros(n, np.float32)
    tmax = np.full(n, 1e5, np.float32)
    # a share of short rays (shadow-ray like) and rays with a positive tmin
    k = n // 8
    tmax[:k] = (rng.random(k) * 5.0).astype(np.float32)
    tmin[k:2 * k] = (rng.random(k) * 2.0).astype(np.float32)
    return o, d, tmin, tmax


@pytest.fixture(scope="module")
def cornell(built):
    sc = scenes.cornell(128, 128)
    tree = sc.initKdTree()
    return sc, tree


@pytest.fixture(scope="module")
def small_bunny(built):
    sc = scenes.small_bunny(128, 128)
    tree = sc.initKdTree()
    return sc, tree


@pytest.fixture(scope="module")
def veach(built):
    sc = scenes.veach_mi(160, 90)
    tree = sc.initKdTree()
    return sc, tree


@pytest.mark.parametrize("layout", [0, 1])
@pytest.mark.parametrize("any_hit", [False, True])
def test_cornell(cornell, layout, any_hit):
    sc, tree = cornell
    flat = tree.flatten(layout)
    dev = DeviceScene(flat)
    o, d, tmin, tmax = _rays(flat, 128, 128, 40000, 3)
    hits = _compare(flat, dev, o, d, tmin, tmax, any_hit)
    assert hits > 10000
    dev.close()


@pytest.mark.parametrize("layout", [0, 1])
@pytest.mark.parametrize("any_hit", [False, True])
def test_small_bunny_with_plane(small_bunny, layout, any_hit):
    sc, tree = small_bunny
    flat = tree.flatten(layout)
    dev = DeviceScene(flat)
    o, d, tmin, tmax = _rays(flat, 128, 128, 60000, 5)
    hits = _compare(flat, dev, o, d, tmin, tmax, any_hit)
    assert hits > 5000
    dev.close()


@pytest.mark.parametrize("any_hit", [False, True])
def test_veach_spheres(veach, any_hit):
    sc, tree = veach
    flat = tree.flatten(1)
    dev = DeviceScene(flat)
    o, d, tmin, tmax = _rays(flat, 160, 90, 60000, 9)
    hits = _compare(flat, dev, o, d, tmin, tmax, any_hit)
    assert hits > 5000
    dev.close()


def test_barycentrics_and_stats(small_bunny):
    sc, tree = small_bunny
    flat = tree.flatten(1)
    dev = DeviceScene(flat)
    o, d, tmin, tmax = _rays(flat, 128, 128, 20000, 11)
    want, ws = orc.oracle_intersect(flat.desc, o, d, tmin, tmax, False, stats=True)
    prim, t, u, v, gs = dev.intersect(o, d, tmin, tmax, stats=True)
    # triangles of this scene carry default texture coordinates (0,0) for every corner, so the
    # oracle's Intersection::u,v are 0; compare barycentrics through the hit position instead:
    # position = o + t*d is identical iff t is, and u,v must reproduce it on the triangle.
    hit = prim >= 0
    tri = hit & (np.ctypeslib.as_array(flat.desc.contents.obj_kind, shape=(sc.n_objects,))[np.maximum(prim, 0)] == 3)
    p = np.ctypeslib.as_array(flat.desc.contents.tri_p, shape=(sc.n_objects, 9))[prim[tri]]
    p1, p2, p3 = p[:, 0:3], p[:, 3:6], p[:, 6:9]
    uu, vv = u[tri][:, None].astype(np.float64), v[tri][:, None].astype(np.float64)
    recon = p1 * (1 - uu - vv) + p2 * uu + p3 * vv
    pos = o[tri].astype(np.float64) + t[tri][:, None].astype(np.float64) * d[tri].astype(np.float64)
    scale = np.maximum(1.0, np.abs(pos).max(axis=1))
    assert np.max(np.abs(recon - pos).max(axis=1) / scale) < 1e-4
    assert (gs.inner, gs.leaves, gs.refs, gs.prim_tests) == (ws.inner, ws.leaves, ws.refs, ws.prim_tests)
    dev.close()


def test_empty_and_errors(cornell):
    sc, tree = cornell
    flat = tree.flatten(1)
    dev = DeviceScene(flat)
    z3 = np.zeros((0, 3), np.float32)
    z1 = np.zeros(0, np.float32)
    prim, t, u, v = dev.intersect(z3, z3, z1, z1)
    assert prim.shape == (0,)
    dev.close()


def test_device_buffers(small_bunny):
    import torch
    sc, tree = small_bunny
    flat = tree.flatten(1)
    dev = DeviceScene(flat)
    o, d, tmin, tmax = _rays(flat, 128, 128, 20000, 13)
    want = orc.oracle_intersect(flat.desc, o, d, tmin, tmax, False)
    to = lambda a: torch.from_numpy(a).cuda()
    prim, t, u, v = dev.intersect(to(o), to(d), to(tmin), to(tmax))
    torch.cuda.synchronize()
    assert np.array_equal(prim.cpu().numpy(), want["prim"])
    hit = want["prim"] >= 0
    assert np.array_equal(_bits(t.cpu().numpy()[hit]), _bits(want["t"][hit]))
    dev.close()


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 511, 4097])
def test_ragged_batch_sizes(small_bunny, n):
    """The ray list is dealt to the waves in 64-ray chunks and lanes are refilled in batches:
    every list length, including those that leave most lanes of the last wave without a ray,
    must produce exactly the per-ray results of the oracle."""
    sc, tree = small_bunny
    flat = tree.flatten(1)
    dev = DeviceScene(flat)
    o, d, tmin, tmax = _rays(flat, 64, 64, 3000, 21)
    rng = np.random.default_rng(n)
    pick = rng.choice(len(o), n, replace=False)
    for any_hit in (False, True):
        _compare(flat, dev, o[pick], d[pick], tmin[pick], tmax[pick], any_hit)
    dev.close()


def _many_transform_scene(n_spheres=150):
    """More transforms than the traversal kernels stage in LDS (64): spheres and a few triangles,
    each with its own scale/rotate/translate, over two planes."""
    from raytracer_amd.host import Scene, Transform
    rng = np.random.default_rng(5)
    sc = Scene()
    m = sc.CreateMaterial()
    glow = sc.CreateMaterial(emission=(1, 1, 1))
    for i in range(n_spheres):
        t = Transform().scale(*(0.15 + 0.3 * rng.random(3))).rotate((0, 1, 0), float(rng.random() * 90))
        t.translate(tuple(float(x) for x in (rng.random(3) * 8 - 4)))
        sc.add_sphere(t, glow if i % 7 == 0 else m)
    for i in range(40):
        t = Transform().rotate((1, 0, 0), float(rng.random() * 180)).translate(tuple(float(x) for x in (rng.random(3) * 8 - 4)))
        p = (rng.random(9) - 0.5).astype(np.float32)
        sc.add_triangle(p, np.tile(np.array([0, 0, 1], np.float32), 3), (0, 0, 1), np.zeros(6, np.float32), t, m)
    sc.add_plane(Transform().scale(9, 9, 1).rotate((1, 0, 0), -90).translate((0, -4.5, 0)), m)
    sc.add_plane(Transform().scale(9, 9, 1).translate((0, 0, -5)), m)
    sc.setSimpleCamera((0, 0, 12), (0, 0, 0), (0, 1, 0), 60.0, 96, 96, 0.0, 10.0)
    return sc


@pytest.mark.parametrize("any_hit", [False, True])
def test_many_transforms_two_planes(built, any_hit):
    sc = _many_transform_scene()
    tree = sc.initKdTree()
    flat = tree.flatten(1)
    assert flat.desc.contents.n_transforms > 64 and flat.desc.contents.n_out_of_tree == 2
    dev = DeviceScene(flat)
    o, d, tmin, tmax = _rays(flat, 96, 96, 40000, 31)
    # rays that start outside the root box and point away from it must come back as misses or plane hits
    far = np.tile(np.array([[50.0, 50.0, 50.0]], np.float32), (64, 1))
    away = np.tile(np.array([[0.577, 0.577, 0.577]], np.float32), (64, 1))
    o, d = np.concatenate([o, far]), np.concatenate([d, away])
    tmin, tmax = np.concatenate([tmin, np.zeros(64, np.float32)]), np.concatenate([tmax, np.full(64, 1e5, np.float32)])
    hits = _compare(flat, dev, o, d, tmin, tmax, any_hit)
    assert hits > 10000
    want, ws = orc.oracle_intersect(flat.desc, o, d, tmin, tmax, any_hit, stats=True)
    gs = dev.intersect(o, d, tmin, tmax, any_hit=any_hit, stats=True)[4]
    assert (gs.inner, gs.leaves, gs.refs, gs.prim_tests) == (ws.inner, ws.leaves, ws.refs, ws.prim_tests)
    dev.close()


def test_nan_rays_do_not_stall(small_bunny):
    """A ray with a NaN component can hit nothing in the tree (every comparison fails); the
    reference finds that out by walking all of it, the kernels skip the walk.  Whatever the planes
    report for such a ray is kept as the reference computes it."""
    sc, tree = small_bunny
    flat = tree.flatten(1)
    dev = DeviceScene(flat)
    o, d, tmin, tmax = _rays(flat, 32, 32, 1000, 41)
    o, d = o.copy(), d.copy()
    o[::7, 1] = np.nan
    d[3::11, 0] = np.nan
    want = orc.oracle_intersect(flat.desc, o, d, tmin, tmax, False)
    prim, t, u, v = dev.intersect(o, d, tmin, tmax)
    assert np.array_equal(prim, want["prim"])
    dev.close()


def test_million_triangle_scene(built):
    """BASELINE configs[2] size class: the stanford bunny subdivided twice (1.1 M triangles,
    26 M nodes, 64 M leaf references, depth 53): table offsets beyond 2^28 bytes, a working set
    larger than the on-chip caches.  Rays and a small frame against the oracle."""
    from raytracer_amd.host import render_params
    sc = scenes.million_bunny(256, 144, env=None)
    sc.add_ibl_pixels(scenes.synthetic_env())
    tree = sc.initKdTree()
    nn, nr, depth = tree.counts()
    assert nn > 20_000_000 and nr > 50_000_000 and depth <= 62
    flat = tree.flatten(1)
    dev = DeviceScene(flat)
    o, d, tmin, tmax = _rays(flat, 256, 144, 60000, 51)
    for any_hit in (False, True):
        hits = _compare(flat, dev, o, d, tmin, tmax, any_hit)
        assert hits > 20000
    p = render_params(256, 144, 2, 2, max_depth=4, seed=5)
    gpu = np.zeros((144, 256, 3), np.float32)
    dev.render(p, gpu)
    cpu = orc.oracle_render(flat.desc, p, 0, threads=16)[0]
    rel_mse = float(np.mean((gpu / 4 - cpu / 4) ** 2 / ((cpu / 4) ** 2 + 1e-2)))
    assert rel_mse <= 1e-5, rel_mse
    dev.close()


def test_successive_calls_on_different_streams_are_ordered(small_bunny):
    """Two asynchronous calls on one handle, each on its own stream, share the handle's chunk-deal
    counters: the library orders them, so every batch comes out complete and identical to a
    one-stream run.  Inputs and outputs are allocated up front and the C-ABI is called directly, so
    that nothing but the library's own scratch is shared between the streams."""
    import ctypes as C
    import torch
    from raytracer_amd import abi
    sc, tree = small_bunny
    flat = tree.flatten(1)
    dev = DeviceScene(flat)
    o, d, tmin, tmax = _rays(flat, 128, 128, 200000, 9)
    n = len(o)
    cols = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (o[:, 0], o[:, 1], o[:, 2], d[:, 0], d[:, 1], d[:, 2], tmin, tmax)]
    rays = abi.Rays(*[c.data_ptr() for c in cols])

    def outputs():
        return [torch.full((n,), -7, dtype=torch.int32, device="cuda")] + [torch.full((n,), -7.0, device="cuda") for _ in range(3)]

    def call(out, flags, stream):
        hits = abi.Hits(*[t.data_ptr() for t in out])
        st = abi.hip().mrt_intersect(dev._h, n, C.byref(rays), C.byref(hits), flags, abi.MRT_BUF_DEVICE, stream, None)
        assert st == abi.MRT_OK, abi.hip().mrt_last_error().decode()

    want, want_any = outputs(), outputs()
    torch.cuda.synchronize()
    call(want, abi.MRT_CLOSEST_HIT, None)
    torch.cuda.synchronize()
    call(want_any, abi.MRT_ANY_HIT, None)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(3):
        a, b, c = outputs(), outputs(), outputs()
        torch.cuda.synchronize()
        call(a, abi.MRT_CLOSEST_HIT, s1.cuda_stream)
        call(b, abi.MRT_ANY_HIT, s2.cuda_stream)
        call(c, abi.MRT_CLOSEST_HIT, s1.cuda_stream)
        torch.cuda.synchronize()
        for x in (a, c):
            for g, w in zip(x, want):
                assert torch.equal(g, w)
        for g, w in zip(b, want_any):
            assert torch.equal(g, w)
    dev.close()


def test_device_probe_reports_the_xcd_mapping(cornell):
    """An unpartitioned MI355X places the workgroups of a grid on its 8 XCDs round-robin; the probe
    made at scene creation must find that (mask 7) or another power-of-two rule -- the traversal
    kernels start each wave on its own XCD's part of the ray list."""
    import ctypes as C
    from raytracer_amd import abi
    sc, tree = cornell
    dev = DeviceScene(tree.flatten(1))
    info = abi.DeviceInfo()
    assert abi.hip().mrt_scene_device_info(dev._h, C.byref(info)) == abi.MRT_OK
    assert info.compute_units > 0 and info.xcd_mask in (0, 1, 3, 7)
    if info.compute_units == 256:
        assert info.xcd_mask == 7
    dev.close()


def test_tree_deeper_than_sixty_two_levels():
    """The reference's SAH build has no depth limit (kdtree.cpp:392-464); the traversal stack takes
    trees up to depth 126.  A hand-made description: 100 thin triangles side by side along x, a
    chain of inner nodes that each split one leaf off to the left, so leaf k sits at depth k + 1."""
    import ctypes as C
    from raytracer_amd import abi
    n = 100
    rng = np.random.default_rng(5)
    tri_p = np.zeros((n, 9), np.float32)
    for k in range(n):
        x0 = np.float32(k)
        tri_p[k] = [x0 + 0.1, -1, 0, x0 + 0.9, -1, 0.2, x0 + 0.5, 1, -0.1]
    tri_n = np.tile(np.array([0, 0, 1] * 3, np.float32), (n, 1))
    tri_uv = np.zeros((n, 6), np.float32)
    sphere = np.zeros((n, 4), np.float32)
    kind = np.full(n, abi.MRT_TRIANGLE, np.uint8)
    mat_idx = np.zeros(n, np.uint32)
    xf_idx = np.zeros(n, np.uint32)
    # node 0 = root; inner node i has children (2i+1: leaf i, 2i+2: the rest); the last "rest" is leaf n-1
    nodes = np.zeros((2 * n - 1, 2), np.uint32)
    refs = np.arange(n, dtype=np.uint32)
    at = 0
    for k in range(n - 1):
        left = 2 * k + 1
        nodes[at, 0] = np.float32(k + 1).view(np.uint32)  # split at x = k + 1
        nodes[at, 1] = 0 | (left << 2)
        nodes[left] = (k, 3 | (1 << 2))
        at = left + 1
    nodes[at] = (n - 1, 3 | (1 << 2))
    d = abi.SceneDesc()
    d.struct_size = C.sizeof(abi.SceneDesc)
    d.n_objects = n
    keep = [kind, mat_idx, xf_idx, tri_p, tri_n, tri_uv, sphere, nodes, refs]
    d.obj_kind, d.obj_material, d.obj_transform = kind.ctypes.data_as(abi.u8p), mat_idx.ctypes.data_as(abi.u32p), xf_idx.ctypes.data_as(abi.u32p)
    d.tri_p, d.tri_n, d.tri_uv, d.sphere = [a.ctypes.data_as(abi.f32p) for a in (tri_p, tri_n, tri_uv, sphere)]
    ident = abi.Transform()
    ident.m[:] = [1, 0, 0, 0, 1, 0, 0, 0, 1]
    ident.inv[:] = [1, 0, 0, 0, 1, 0, 0, 0, 1]
    d.n_transforms, d.transforms = 1, C.pointer(ident)
    m = abi.Material()
    m.albedo[:] = [0.5, 0.5, 0.5]
    m.ior, m.roughness, m.metalness, m.type, m.texture = 1.0, 0.5, 0.0, 0, -1
    d.n_materials, d.materials = 1, C.pointer(m)
    d.n_nodes, d.nodes = len(nodes), C.cast(nodes.ctypes.data, C.POINTER(abi.KdNode))
    d.n_refs, d.refs = n, refs.ctypes.data_as(abi.u32p)
    d.root_min[:] = [-1, -2, -1.2]
    d.root_max[:] = [n + 1, 2, 1.2]
    d.env_light = -1
    d.env_inv[:] = [1, 0, 0, 0, 1, 0, 0, 0, 1]
    h = C.c_void_p()
    st = abi.hip().mrt_scene_create(C.byref(d), 0, C.byref(h))
    assert st == abi.MRT_OK, abi.hip().mrt_last_error().decode()
    info = abi.DeviceInfo()
    abi.hip().mrt_scene_device_info(h, C.byref(info))
    assert info.kd_tree_depth == n - 1
    # rays along +x / -x through all the triangles' slabs (they defer a node at every level), and
    # random rays
    m_rays = 20000
    o = np.zeros((m_rays, 3), np.float32)
    dr = np.zeros((m_rays, 3), np.float32)
    o[:, 0] = np.where(np.arange(m_rays) % 2 == 0, -0.5, n + 0.5)
    o[:, 1] = rng.random(m_rays, dtype=np.float32) * 1.6 - 0.8
    o[:, 2] = rng.random(m_rays, dtype=np.float32) * 0.6 - 0.3
    tgt = np.stack([rng.random(m_rays, dtype=np.float32) * n, rng.random(m_rays, dtype=np.float32) * 1.6 - 0.8,
                    rng.random(m_rays, dtype=np.float32) * 0.4 - 0.2], axis=1).astype(np.float32)
    dr = tgt - o
    dr = (dr / np.linalg.norm(dr, axis=1, keepdims=True)).astype(np.float32)
    tmin, tmax = np.zeros(m_rays, np.float32), np.full(m_rays, 1e5, np.float32)
    for any_hit in (False, True):
        want = orc.oracle_intersect(C.pointer(d), o, dr, tmin, tmax, any_hit)
        cols = [np.ascontiguousarray(o[:, k]) for k in range(3)] + [np.ascontiguousarray(dr[:, k]) for k in range(3)] + [tmin, tmax]
        prim, t, u, v = np.empty(m_rays, np.int32), np.empty(m_rays, np.float32), np.empty(m_rays, np.float32), np.empty(m_rays, np.float32)
        rays = abi.Rays(*[c.ctypes.data for c in cols])
        hits = abi.Hits(prim.ctypes.data, t.ctypes.data, u.ctypes.data, v.ctypes.data)
        st = abi.hip().mrt_intersect(h, m_rays, C.byref(rays), C.byref(hits), abi.MRT_ANY_HIT if any_hit else abi.MRT_CLOSEST_HIT, abi.MRT_BUF_HOST, None, None)
        assert st == abi.MRT_OK
        assert np.array_equal(prim, want["prim"])
        hit = prim >= 0
        assert hit.sum() > 2000 and len(np.unique(prim[hit])) > 60
        assert np.array_equal(_bits(t[hit]), _bits(want["t"][hit]))
    abi.hip().mrt_scene_destroy(h)
    del keep


def test_object_order_inside_the_library_is_invisible(built, monkeypatch):
    """mrt_scene_create keeps the objects in the order in which the leaves first reference them; the
    caller's indices come back from mrt_intersect.  A mesh whose objects were shuffled (so the two
    orders differ everywhere), spheres, a plane, transforms: hits equal the oracle's on the caller's
    description, and equal bit for bit what the library returns with the caller's order kept
    (MRT_OBJECT_ORDER=input); so do the rendered frames."""
    from raytracer_amd import abi
    from raytracer_amd.host import render_params
    sc = _many_transform_scene()
    abi.host().mrth_shuffle_objects(sc._h, 99)
    sc.add_ibl_pixels(scenes.synthetic_env(), also_env=True)
    flat = sc.initKdTree().flatten(1)
    o, d, tmin, tmax = _rays(flat, 96, 96, 30000, 41)
    out = []
    for keep in (False, True):
        if keep:
            monkeypatch.setenv("MRT_OBJECT_ORDER", "input")
        else:
            monkeypatch.delenv("MRT_OBJECT_ORDER", raising=False)
        dev = DeviceScene(flat)
        for any_hit in (False, True):
            assert _compare(flat, dev, o, d, tmin, tmax, any_hit) > 5000
        hits = dev.intersect(o, d, tmin, tmax, any_hit=False)
        img = np.zeros((48, 64, 3), np.float32)
        dev.render(render_params(64, 48, 2, 2, max_depth=4, seed=5), img)
        out.append((hits, img))
        dev.close()
    for a, b in zip(out[0][0], out[1][0]):
        assert np.array_equal(_bits(a) if a.dtype == np.float32 else a, _bits(b) if b.dtype == np.float32 else b)
    assert np.array_equal(out[0][1], out[1][1]) and out[0][1].max() > 0


@pytest.mark.parametrize("name", ["cornell", "bunny_plane", "transforms", "fuzz3", "fuzz6"])
def test_where_the_leaf_lists_live_is_invisible(built, monkeypatch, name):
    """The leaves' reference lists inside the pool of tree records right behind the pair of records that
    holds them, groups that fit a cache line kept from straddling two (the default), the same packed without
    regard to lines (MRT_REFS_INLINE=1), or in an array of their own (=0): hits equal the oracle's, every output array
    and the traversal counters are the same bits in all three, both node layouts, packets included."""
    from tests import fuzz_scenes
    if name == "cornell":
        sc = scenes.cornell(96, 96)
    elif name == "bunny_plane":
        sc = scenes.small_bunny(96, 96, env_pixels=scenes.synthetic_env())
    elif name == "transforms":
        sc = _many_transform_scene()
    else:
        sc = fuzz_scenes.make(int(name[4:]))  # leaves of one, many and no primitives, ties
    tree = sc.initKdTree(2 if name.startswith("fuzz") else 16)
    for layout in (0, 1):
        flat = tree.flatten(layout)
        o, d, tmin, tmax = _rays(flat, 96, 96, 20000, 5)
        out = []
        for mode in (None, "0", "1"):
            if mode is None:
                monkeypatch.delenv("MRT_REFS_INLINE", raising=False)
            else:
                monkeypatch.setenv("MRT_REFS_INLINE", mode)
            dev = DeviceScene(flat)
            for any_hit in (False, True):
                assert _compare(flat, dev, o, d, tmin, tmax, any_hit) > 1000
            res = dev.intersect(o, d, tmin, tmax, stats=True)
            monkeypatch.setenv("MRT_INTERSECT_PACKETS", "1")
            pk = DeviceScene(flat)
            monkeypatch.delenv("MRT_INTERSECT_PACKETS")
            res_pk = pk.intersect(o, d, tmin, tmax)
            pk.close()
            dev.close()
            gs = res[4]
            out.append((res[:4], (gs.inner, gs.leaves, gs.refs, gs.prim_tests), res_pk))
        for other in out[1:]:
            assert other[1] == out[0][1]
            for a, b in zip(out[0][0] + out[0][2], other[0] + other[2]):
                assert np.array_equal(_bits(a) if a.dtype == np.float32 else a, _bits(b) if b.dtype == np.float32 else b)


@pytest.mark.parametrize("seed", list(range(12)))
def test_random_scenes_with_ties_against_the_oracle(built, seed):
    """Differential test on seeded random scenes built to produce ties (tests/fuzz_scenes.py):
    mrt_intersect against the oracle, closest and shadow rays, both node layouts, hit ids, distances
    and barycentrics bit for bit, and the traversal counters."""
    from tests import fuzz_scenes
    sc = fuzz_scenes.make(seed)
    tree = sc.initKdTree(1 + seed % 4)
    for layout in (0, 1):
        flat = tree.flatten(layout)
        dev = DeviceScene(flat)
        o, d, tmin, tmax = fuzz_scenes.rays(flat, seed)
        o2, d2, tmin2, tmax2 = fuzz_scenes.rays_hard(flat, seed)  # from vertices, along edges, -0.0 components, empty intervals
        o, d, tmin, tmax = np.concatenate([o, o2]), np.concatenate([d, d2]), np.concatenate([tmin, tmin2]), np.concatenate([tmax, tmax2])
        for any_hit in (False, True):
            assert _compare(flat, dev, o, d, tmin, tmax, any_hit) > 500
        want, ws = orc.oracle_intersect(flat.desc, o, d, tmin, tmax, False, stats=True)
        prim, t, u, v, gs = dev.intersect(o, d, tmin, tmax, stats=True)
        kinds = np.ctypeslib.as_array(flat.desc.contents.obj_kind, shape=(sc.n_objects,))
        tri = (prim >= 0) & (kinds[np.maximum(prim, 0)] == 3)  # the scenes' texture coordinates make the reference's u, v the barycentrics
        assert tri.sum() > 300
        # equal as numbers: the reference's interpolation u * 1 + v * 0 + w * 0 turns a barycentric of -0 into +0
        assert np.array_equal(u[tri], want["u"][tri]) and np.array_equal(v[tri], want["v"][tri])
        assert (gs.inner, gs.leaves, gs.refs, gs.prim_tests) == (ws.inner, ws.leaves, ws.refs, ws.prim_tests)
        dev.close()
