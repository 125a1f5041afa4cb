"""Seeded random scenes for differential tests of kd-tree build + traversal: the constructions that
stress tie-breaking and the sweep of the SAH builder -- vertices on a coarse grid (many equal event
positions of mixed kinds), axis-parallel flat triangles, exact duplicates and coplanar overlaps (equal
hit distances: the first-listed primitive must win), slivers below the |det| < 1e-4 rejection,
overlapping spheres under anisotropic transforms, emissive primitives (shadow rays skip them but
shrink the interval), planes outside the tree."""
import numpy as np

from raytracer_amd.host import Scene, Transform


def make(seed):
    rng = np.random.default_rng(seed)
    sc = Scene()
    dull = sc.CreateMaterial()
    glow = sc.CreateMaterial(emission=(2, 2, 2))
    ident = Transform()
    moved = Transform().scale(1.5, 0.7, 1.1).rotate((0.3, 0.9, 0.1), 25.0).translate((0.5, -0.25, 0.125))
    grid = int(rng.integers(3, 9))  # coordinates are multiples of 1/grid: shared planes everywhere
    n_tri = int(rng.integers(40, 400))
    fn, uv = (0, 0, 1), np.array([0, 0, 1, 0, 0, 1], np.float32)  # texture coordinates that make Intersection::u, v the barycentrics (triangle.cpp:46-49)
    nrm = np.tile(np.array([0, 0, 1], np.float32), 3)
    tris = []
    for i in range(n_tri):
        kind = rng.integers(0, 10)
        if kind < 5:      # grid-aligned vertices
            p = rng.integers(-2 * grid, 2 * grid + 1, 9).astype(np.float32) / np.float32(grid)
        elif kind < 7:    # flat in one axis
            p = (rng.random(9) * 4 - 2).astype(np.float32)
            ax = int(rng.integers(0, 3))
            p[ax::3] = np.float32(rng.integers(-2 * grid, 2 * grid + 1)) / np.float32(grid)
        elif kind < 8 and tris:  # an exact duplicate of an earlier triangle
            p = tris[int(rng.integers(0, len(tris)))].copy()
        elif kind < 9 and tris:  # coplanar with an earlier one, overlapping it
            q = tris[int(rng.integers(0, len(tris)))]
            w = rng.random((3, 3)).astype(np.float32)
            w /= w.sum(axis=1, keepdims=True)
            p = (w @ q.reshape(3, 3)).reshape(9).astype(np.float32) * np.float32(1.5)
        else:             # a sliver
            a = (rng.random(3) * 4 - 2).astype(np.float32)
            p = np.concatenate([a, a + np.float32(1e-3) * rng.random(3).astype(np.float32), a + (rng.random(3) - 0.5).astype(np.float32)])
        tris.append(p.astype(np.float32))
        sc.add_triangle(p, nrm, fn, uv, moved if i % 5 == 0 else ident, glow if i % 11 == 0 else dull)
    for i in range(int(rng.integers(0, 12))):
        t = Transform().scale(*(0.2 + rng.random(3))).rotate((0, 1, 0), float(rng.random() * 90)).translate(tuple(float(x) for x in (rng.random(3) * 3 - 1.5)))
        sc.add_sphere(t, glow if i % 4 == 0 else dull)
    for i in range(int(rng.integers(0, 3))):
        sc.add_plane(Transform().scale(5, 5, 1).rotate((1, 0, 0), -90.0 + 30.0 * i).translate((0, -2.5 - i, 0)), dull)
    sc.setSimpleCamera((0.3, 0.4, 7.0), (0, 0, 0), (0, 1, 0), 45.0, 64, 64, 0.0, 7.0)
    return sc


def rays(flat, seed, n_random=6000):
    """Camera rays, rays between random points of the root box, and axis-parallel rays through grid
    coordinates (they run inside split planes and along triangle edges)."""
    from tests import orc
    rng = np.random.default_rng(seed + 1000)
    o1, d1 = orc.camera_rays(flat.desc, 48, 48, seed=seed)
    o2, d2 = orc.random_rays(flat.desc, n_random, seed=seed + 1)
    k = 1500
    o3 = (rng.integers(-16, 17, (k, 3)) / 8.0).astype(np.float32)
    d3 = np.zeros((k, 3), np.float32)
    d3[np.arange(k), rng.integers(0, 3, k)] = rng.choice(np.array([-1.0, 1.0], np.float32), k)
    o3 -= d3 * np.float32(4.0)
    o, d = np.concatenate([o1, o2, o3]), np.concatenate([d1, d2, d3])
    n = len(o)
    tmin, tmax = np.zeros(n, np.float32), np.full(n, 1e5, np.float32)
    tmax[::7] = (rng.random(len(tmax[::7])) * 6).astype(np.float32)
    tmin[3::9] = (rng.random(len(tmin[3::9])) * 2).astype(np.float32)
    return np.ascontiguousarray(o), np.ascontiguousarray(d), tmin, tmax


def rays_hard(flat, seed, n=12000):
    """Rays aimed at the places where ties and degenerate arithmetic live: from vertices, edge
    midpoints and centroids of the scene's (untransformed) triangles, towards other vertices (through
    shared vertices and along edges), along the axes with -0.0 components, with empty, zero-length and
    inverted intervals."""
    rng = np.random.default_rng(seed + 5000)
    d_ = flat.desc.contents
    m = d_.n_objects
    kinds = np.ctypeslib.as_array(d_.obj_kind, shape=(m,))
    xf = np.ctypeslib.as_array(d_.obj_transform, shape=(m,))
    tri = np.ctypeslib.as_array(d_.tri_p, shape=(m, 9))[(kinds == 3) & (xf == 0)].reshape(-1, 3, 3).astype(np.float32)
    verts = tri.reshape(-1, 3)
    mids = ((tri[:, 0] + tri[:, 1]) * np.float32(0.5))
    cents = ((tri[:, 0] + tri[:, 1] + tri[:, 2]) / np.float32(3))
    pts = np.concatenate([verts, mids, cents]).astype(np.float32)
    a = pts[rng.integers(0, len(pts), n)]
    b = pts[rng.integers(0, len(pts), n)]
    d = (b - a).astype(np.float32)
    axis = np.zeros((n, 3), np.float32)
    axis[np.arange(n), rng.integers(0, 3, n)] = rng.choice(np.array([-1.0, 1.0], np.float32), n)
    use_axis = rng.random(n) < 0.35
    d[use_axis] = axis[use_axis]
    d[(d == 0).all(axis=1)] = np.float32([0, 0, 1])
    neg0 = rng.random((n, 3)) < 0.3
    d = np.where((d == 0) & neg0, np.float32(-0.0), d).astype(np.float32)
    back = rng.random(n) < 0.5            # half of the rays start before the point and pass through it
    o = np.where(back[:, None], a - d * np.float32(0.5), a).astype(np.float32)
    tmin, tmax = np.zeros(n, np.float32), np.full(n, 1e5, np.float32)
    k = rng.random(n)
    tmax[k < 0.1] = 0.0
    tmax[(k >= 0.1) & (k < 0.25)] = (rng.random(int(((k >= 0.1) & (k < 0.25)).sum())) * 1.5).astype(np.float32)
    tmin[(k >= 0.25) & (k < 0.35)] = (rng.random(int(((k >= 0.25) & (k < 0.35)).sum())) * 1.5).astype(np.float32)
    inv = (k >= 0.35) & (k < 0.4)
    tmin[inv], tmax[inv] = 2.0, 1.0
    return np.ascontiguousarray(o), np.ascontiguousarray(d), tmin, tmax


def make_shaded(seed, min_roughness=0.0, rich=False):
    """A random scene for the shading loop: meshes of random Cook-Torrance materials (all three types,
    roughness down to 0, checker textures), spheres, a floor plane, and lights drawn from: a synthetic
    environment map, sphere lights, triangle lights; thin-lens camera.  rich: also image textures under
    rotated and scaled texture transforms, random texture coordinates and smooth (unnormalised) vertex
    normals, and a third of the triangles under rotated, non-uniformly scaled object transforms."""
    from raytracer_amd import abi, scenes
    rng = np.random.default_rng(seed + 9000)
    sc = Scene()
    mats = []
    for i in range(int(rng.integers(3, 9))):
        typ = int(rng.choice([abi.MRT_MAT_DIFFUSE, abi.MRT_MAT_SPECULAR, abi.MRT_MAT_TRANSPARENT], p=[0.5, 0.3, 0.2]))
        rough = max(float(rng.choice([0.0, 0.002, 0.05, 0.3, 0.8, 1.0])), min_roughness)
        m = sc.CreateMaterial(typ, albedo=tuple(float(x) for x in rng.random(3)), roughness=rough, metalness=float(rng.choice([0.0, 0.1, 1.0])),
                              ior=float(1.0 + rng.random()))
        if typ != abi.MRT_MAT_TRANSPARENT and rng.random() < 0.3:
            if rich and rng.random() < 0.5:
                tw, th = int(rng.integers(1, 9)), int(rng.integers(1, 9))
                txf = Transform().scale(float(0.2 + rng.random()), float(0.2 + rng.random()), 1).rotate((0, 0, 1), float(rng.random() * 360)).translate(
                    (float(rng.random()), float(rng.random()), 0))
                sc.set_image_texture(m, (rng.integers(0, 256, (th, tw, 3)) / 255.0).astype(np.float32), txf)
            else:
                sc.set_checker(m, tuple(float(x) for x in rng.random(3)), tuple(float(x) for x in rng.random(3)), Transform().scale(0.25, 0.25, 1))
        mats.append(m)
    xfs = [Transform()]
    if rich:
        for i in range(4):
            xfs.append(Transform().scale(*(float(x) for x in 0.5 + rng.random(3))).rotate(tuple(float(x) for x in rng.random(3) + 0.1), float(rng.random() * 360)).translate(
                tuple(float(x) for x in rng.random(3) - 0.5)))
    nrm = np.tile(np.array([0, 0, 1], np.float32), 3)
    uv = np.array([0, 0, 1, 0, 0, 1], np.float32)
    for i in range(int(rng.integers(30, 200))):
        c = (rng.random(3) * 4 - 2).astype(np.float32)
        p = (np.tile(c, 3) + (rng.random(9) - 0.5).astype(np.float32) * np.float32(1.6)).astype(np.float32)
        e1, e2 = p[3:6] - p[0:3], p[6:9] - p[0:3]
        fn = np.cross(e1, e2)
        fn = fn / max(np.linalg.norm(fn), 1e-20)
        n3 = np.tile(fn.astype(np.float32), 3) if rng.random() < 0.7 else nrm
        tuv, xf = uv, xfs[0]
        if rich:
            if rng.random() < 0.4:
                n3 = (n3 + (rng.random(9) - 0.5).astype(np.float32) * np.float32(0.8)).astype(np.float32)
            tuv = (rng.random(6) * 3 - 1).astype(np.float32)
            xf = xfs[int(rng.integers(0, len(xfs)))] if rng.random() < 0.35 else xfs[0]
        sc.add_triangle(p, n3, tuple(float(x) for x in fn), tuv, xf, mats[int(rng.integers(0, len(mats)))])
    for i in range(int(rng.integers(1, 6))):
        t = Transform().scale(*(0.3 + 0.5 * rng.random(3))).translate(tuple(float(x) for x in (rng.random(3) * 3 - 1.5)))
        sc.add_sphere(t, mats[int(rng.integers(0, len(mats)))])
    sc.add_plane(Transform().scale(6, 6, 1).rotate((1, 0, 0), -90.0).translate((0, -2.2, 0)), mats[0])
    lights = int(rng.integers(1, 8))  # bit 0: environment, bit 1: sphere lights, bit 2: triangle light
    if lights & 1:
        sc.add_ibl_pixels(scenes.synthetic_env(seed=seed), also_env=True)
    if lights & 2:
        for i in range(int(rng.integers(1, 4))):
            e = float(rng.choice([5.0, 40.0, 300.0]))
            m = sc.CreateMaterial(abi.MRT_MAT_DIFFUSE, albedo=(1, 1, 1), emission=(e, e, e))
            r = float(rng.choice([0.05, 0.2, 0.6]))
            t = Transform().scale(r, r, r).translate(tuple(float(x) for x in (rng.random(3) * 3 - 1.5 + np.array([0, 2.0, 0]))))
            sc.add_sphere(t, m)
            sc.add_sphere_light(t, m, (e, e, e))
    if lights & 4:
        e = 25.0
        m = sc.CreateMaterial(abi.MRT_MAT_DIFFUSE, albedo=(1, 1, 1), emission=(e, e, e))
        p = np.array([-0.6, 2.4, -0.6, 0.6, 2.4, -0.6, 0.0, 2.4, 0.7], np.float32)
        n3 = np.tile(np.array([0, -1, 0], np.float32), 3)
        sc.add_triangle(p, n3, (0, -1, 0), uv, Transform(), m)
        sc.add_triangle_light(p, n3, (0, -1, 0), Transform(), m, (e, e, e))
    sc.setSimpleCamera((0.4, 0.6, 6.5), (0, 0, 0), (0, 1, 0), 50.0, 64, 48, float(rng.choice([0.0, 0.05])), 6.5)
    return sc
