"""Device shading functions (mrt_debug_eval) and mrt_resolve against values recorded from the
reference's own code (tests/golden/*.npz) where the reference builds here, and against the CPU
oracle for Material::eval, whose translation unit does not.

Tolerances: table lookups, comparisons and plain fp32 arithmetic are bit-exact; results that go
through the device's sin / cos / sqrt / pow are compared to a few ulp (stated per check).
"""
import ctypes as C
import os

import numpy as np
import pytest

from raytracer_amd import abi, scenes
from raytracer_amd.host import DeviceScene, Scene, Transform
from tests import orc
from tests.golden import cases

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
P = orc._p
DBG = {"MATERIAL_EVAL": 0, "MATERIAL_SAMPLE": 1, "LIGHT_SAMPLE": 2, "LIGHT_PDF": 3, "IBL": 4, "RANDOM_DIR": 5, "SAMPLER": 6,
       "CAMERA": 7, "BSDF_TERMS": 8, "TEXTURE": 9, "LIGHT_PICK": 10, "RANDOM_DIR_DRAWS": 11}


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def near(a, b, rel, abs_=1e-7, frac=1.0):
    ok = np.abs(a - b) <= abs_ + rel * np.abs(b)
    assert ok.mean() >= frac, "%.5f of the entries within %g (worst %g)" % (ok.mean(), rel, np.abs(a - b).max())


def test_textures_random_dir_light_pick_against_reference_goldens(built):
    z, g = np.load(os.path.join(GOLD, "shading_l2.npz")), cases.shading_inputs()
    n = g["n"]
    sc = cases.texture_scene()
    flat = sc.initKdTree().flatten(1)
    dev = DeviceScene(flat)
    for tex, key in ((0, "checker_value"), (1, "image_value")):
        rows = np.zeros((n, 3), np.float32)
        rows[:, 0], rows[:, 1:3] = tex, g["uv"]
        out = dev.debug_eval(DBG["TEXTURE"], rows, 3)
        assert np.array_equal(bits(out), z[key]), key
    rows = np.zeros((n, 8), np.float32)
    rows[:, 0:3], rows[:, 3:8] = g["normal"], z["random_dir_draws"].view(np.float32)
    out = dev.debug_eval(DBG["RANDOM_DIR_DRAWS"], rows, 3)
    near(out, z["random_dir"].view(np.float32), 4e-6, 2e-7)  # device sinf / cosf against glibc's
    # an out-of-range texture index is refused on the host, not dereferenced on the device
    rows = np.zeros((1, 3), np.float32)
    rows[0, 0] = 2
    with pytest.raises(Exception):
        dev.debug_eval(DBG["TEXTURE"], rows, 3)
    dev.close()
    # UniformLightDistribution: scenes with 1, 2, 3, 7 and 32 lights
    for nl in cases.LIGHT_COUNTS:
        s = Scene()
        m = s.CreateMaterial(abi.MRT_MAT_DIFFUSE, emission=(1, 1, 1))
        for k in range(nl):
            t = Transform().scale(0.1, 0.1, 0.1).translate((0.3 * k, 0.0, 0.0))
            s.add_sphere(t, m)
            s.add_sphere_light(t, m, (1, 1, 1))
        s.setSimpleCamera((0, 0, 5), (0, 0, 0), (0, 1, 0), 40, 16, 16, 0.0, 5.0)
        f = s.initKdTree().flatten(1)
        d = DeviceScene(f)
        out = d.debug_eval(DBG["LIGHT_PICK"], g["u1"].reshape(-1, 1), 2)
        assert np.array_equal(out[:, 0].astype(np.int32), z["uniform_light_off_%d" % nl])
        assert np.array_equal(bits(out[:, 1]), z["uniform_light_pdf_%d" % nl])
        d.close()


def test_camera_against_recorded_reference_rays(built):
    """SimpleCamera::get_ray on the device (MRT_DBG_CAMERA) against the rays recorded from the
    reference's Camera.cpp for the scene-9, scene-20, scene-21 and a pinhole camera.  The camera
    parameters come from the host library's constructor (tan() of the host's libm), hence 5e-6."""
    z, g = np.load(os.path.join(GOLD, "functions.npz")), cases.function_inputs()
    n = g["n"]
    for name, c in cases.CAMERA_CASES.items():
        sc = Scene()
        m = sc.CreateMaterial()
        sc.add_sphere(Transform(), m)
        sc.setSimpleCamera(c[0], c[1], c[2], *[float(x) for x in c[3:]])
        flat = sc.initKdTree().flatten(1)
        dev = DeviceScene(flat)
        rows = np.zeros((n, 4), np.float32)
        rows[:, 0:2], rows[:, 2:4] = g["st"], g["u2"]
        out = dev.debug_eval(DBG["CAMERA"], rows, 6)
        near(out[:, 0:3], z["camera_%s_o" % name].view(np.float32), 5e-6, 1e-6)
        near(out[:, 3:6], z["camera_%s_d" % name].view(np.float32), 5e-6, 1e-6)
        dev.close()


def test_material_eval_against_oracle(built):
    """Material::eval (Material.cpp:983-1013) through MRT_DBG_MATERIAL_EVAL for rough, glossy and
    near-mirror Cook-Torrance materials, a textured one and a dielectric (which evaluates to zero)."""
    rng = np.random.default_rng(17)
    n = 4096
    sc = cases.texture_scene()  # materials 0 (checker), 1 (image)
    mats = [sc.CreateMaterial(abi.MRT_MAT_DIFFUSE, albedo=(0.2, 0.7, 0.4), roughness=0.5, metalness=0.1),
            sc.CreateMaterial(abi.MRT_MAT_SPECULAR, albedo=(1, 1, 1), roughness=0.2, metalness=0.9),
            sc.CreateMaterial(abi.MRT_MAT_SPECULAR, albedo=(0.9, 0.8, 0.7), roughness=0.002, metalness=1.0),
            sc.CreateMaterial(abi.MRT_MAT_TRANSPARENT, albedo=(1, 1, 1), ior=1.5, roughness=0.01)]
    flat = sc.initKdTree().flatten(1)
    dev = DeviceScene(flat)

    def unit(k):
        v = rng.normal(size=(k, 3)).astype(np.float32)
        return np.ascontiguousarray((v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32))

    lib = orc.oracle()
    for mat in mats:
        nrm, rd = unit(n), unit(n)
        flip = (rd * nrm).sum(1) > 0
        rd[flip] = -rd[flip]  # the ray arrives against the normal
        # wi: half of them near the mirror direction (where the specular lobe lives), half anywhere
        refl = rd - 2 * (rd * nrm).sum(1, keepdims=True) * nrm
        wi = unit(n)
        k = n // 2
        wi[:k] = refl[:k] + np.float32(0.05) * wi[:k]
        wi = np.ascontiguousarray((wi / np.linalg.norm(wi, axis=1, keepdims=True)).astype(np.float32))
        rows = np.zeros((n, 10), np.float32)
        rows[:, 0], rows[:, 1:4], rows[:, 4:7], rows[:, 7:10] = mat, rd, nrm, wi
        got = dev.debug_eval(DBG["MATERIAL_EVAL"], rows, 4)
        m = sc.material(mat)
        want_f, want_p = np.zeros((n, 3), np.float32), np.zeros(n, np.float32)
        lib.orc_material_eval(C.byref(m), n, P(rd), P(nrm), P(wi), P(want_f), P(want_p))
        near(got[:, 0:3], want_f, 1e-4, 1e-7, frac=0.999)  # exp / pow of the device against glibc's
        near(got[:, 3], want_p, 1e-4, 1e-7, frac=0.999)
        if sc.material(mat).type == abi.MRT_MAT_TRANSPARENT:
            assert not got.any()
        else:
            assert (got[:, 3] > 0).mean() > 0.1
    dev.close()


def test_light_sampling_on_cornell_triangle_lights(built):
    """Triangle::sample / Object::pdf through a ShapeLight (triangle.cpp:86-104, Object.cpp:13-22)
    on the two emissive triangles the MTL loader turns into lights."""
    rng = np.random.default_rng(23)
    n = 4096
    sc = scenes.cornell(32, 32)
    flat = sc.initKdTree().flatten(1)
    d = flat.desc.contents
    assert d.n_lights == 2 and all(d.lights[i].shape_kind == abi.MRT_TRIANGLE for i in range(2))
    dev = DeviceScene(flat)
    lib = orc.oracle()
    for light in range(2):
        centre = np.array(d.lights[light].tri_p, np.float32).reshape(3, 3).mean(0)
        pos = (centre + (rng.random((n, 3), dtype=np.float32) - 0.5) * np.float32(4.0)).astype(np.float32)
        pos[:, 1] = centre[1] - np.float32(0.2) - rng.random(n, dtype=np.float32) * 3  # below the ceiling light
        v = rng.normal(size=(n, 3)).astype(np.float32)
        nrm = np.ascontiguousarray((v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32))
        u = rng.random((n, 2), dtype=np.float32)
        rows = np.zeros((n, 10), np.float32)
        rows[:, 0], rows[:, 1:4], rows[:, 4:7], rows[:, 7], rows[:, 8:10] = light, pos, nrm, 1, u
        g = dev.debug_eval(DBG["LIGHT_SAMPLE"], rows, 7)
        op, on, opdf = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32), np.zeros(n, np.float32)
        lib.orc_light_shape_sample(flat.desc, light, n, P(pos), P(nrm), P(np.ones(n, np.uint8)), P(u), P(op), P(on), P(opdf))
        near(g[:, 0:3], op, 2e-6, 1e-6)  # sqrtf in UniformSampleTriangle
        assert np.array_equal(bits(g[:, 3:6]), bits(on)) and np.array_equal(bits(g[:, 6]), bits(opdf))
        # directions from the points to sampled points of the light: Object::pdf must find it
        wi = op - pos
        wi = np.ascontiguousarray((wi / np.linalg.norm(wi, axis=1, keepdims=True)).astype(np.float32))
        rows[:, 7:10] = wi
        g = dev.debug_eval(DBG["LIGHT_PDF"], rows, 1)
        lib.orc_light_shape_pdf(flat.desc, light, n, P(pos), P(nrm), P(wi), P(opdf))
        assert (opdf > 0).mean() > 0.5
        near(g[:, 0], opdf, 1e-6, 0.0, frac=0.999)  # a grazing hit on the triangle's edge may flip
    dev.close()


def test_resolve_average_and_gamma(built):
    """mrt_resolve = powf(sum / spp, 1 / 2.2) (CPURenderer.cpp:64-69), host and device buffers."""
    import torch
    sc = scenes.cornell(16, 16)
    flat = sc.initKdTree().flatten(1)
    dev = DeviceScene(flat)
    rng = np.random.default_rng(4)
    n_pix, spp = 5000, 37
    sums = (rng.random((n_pix, 3), dtype=np.float32) * np.float32(80.0)).astype(np.float32)
    sums[:7] = 0
    want = np.power(sums / np.float32(spp), np.float32(1.0 / 2.2), dtype=np.float32)
    out = np.full_like(sums, -1)
    st = abi.hip().mrt_resolve(dev._h, n_pix, sums.ctypes.data, spp, out.ctypes.data, abi.MRT_BUF_HOST, None)
    assert st == abi.MRT_OK
    near(out, want, 2e-6, 1e-7)  # device powf
    assert not out[:7].any()
    t_in = torch.from_numpy(sums).cuda()
    t_out = torch.full_like(t_in, -1)
    st = abi.hip().mrt_resolve(dev._h, n_pix, t_in.data_ptr(), spp, t_out.data_ptr(), abi.MRT_BUF_DEVICE, torch.cuda.current_stream().cuda_stream)
    assert st == abi.MRT_OK
    torch.cuda.synchronize()
    assert np.array_equal(t_out.cpu().numpy(), out)
    # an end-to-end frame: resolve(render) equals the oracle's frame through the same formula
    from raytracer_amd.host import render_params
    p = render_params(16, 16, 2, 2, max_depth=3, seed=5)
    img = np.zeros((16, 16, 3), np.float32)
    dev.render(p, img)
    cpu, _, _, _ = orc.oracle_render(flat.desc, p, 0, threads=4)
    res = np.zeros_like(img)
    assert abi.hip().mrt_resolve(dev._h, 256, img.ctypes.data, 4, res.ctypes.data, abi.MRT_BUF_HOST, None) == abi.MRT_OK
    near(res, np.power(cpu / np.float32(4), np.float32(1.0 / 2.2), dtype=np.float32), 1e-4, 1e-5, frac=0.995)
    dev.close()


def test_device_functions_on_edge_values(built):
    """Zero, -0, 1, denormal, tiny, huge, infinite and NaN arguments (a quarter of every column) through
    the device functions against the oracle, which equals the reference's bsdf.cpp bit for bit on the
    same kind of input (checked live in the container).  A value that is NaN on one side must be NaN
    on the other; arithmetic-only results are bit-exact; results through expf / acosf / atan2f agree
    to the usual 2e-6 relative.  Texture coordinates that are NaN, infinite or beyond the int range
    select the same texel on both sides (texel_index)."""
    import ctypes as C
    lib = orc.oracle()
    vp = C.c_void_p
    lib.orc_texture_value.argtypes = [C.POINTER(abi.SceneDesc), C.c_int, C.c_int64, vp, vp]
    lib.orc_ibl.argtypes = [C.POINTER(abi.SceneDesc), C.c_int64, vp, vp]
    P = orc._p
    rng = np.random.default_rng(99)
    n = 20000
    edge = np.array([0, -0.0, 1, -1, 0.5, 1e-8, 1e-20, 1e-38, 1e-45, 0.002, 4e-6, 0.9999999, 1.0000001, 2, 3e9, -3e9, 1e8, 1e30,
                     np.inf, -np.inf, np.nan], np.float32)

    def draw(lo, hi, k=n):
        a = rng.uniform(lo, hi, k).astype(np.float32)
        m = rng.random(k) < 0.25
        a[m] = edge[rng.integers(0, len(edge), int(m.sum()))]
        return a

    def same_class(a, b):
        assert np.array_equal(np.isnan(a), np.isnan(b))
        assert np.array_equal(np.isinf(a), np.isinf(b))

    def near_or_equal(a, b, rel):
        same_class(a, b)
        f = np.isfinite(b)
        with np.errstate(all="ignore"):
            ok = (bits(a) == bits(b)) | (np.abs(a - b) <= rel * np.abs(b) + 1e-37) | ~f
        assert ok.all(), (a[~ok][:4], b[~ok][:4])

    sc = cases.texture_scene()
    flat = sc.initKdTree().flatten(1)
    dev = DeviceScene(flat)
    ndoth, alpha, ldoth, ext, inn, ddotn = draw(-1, 1), draw(0, 1), draw(-1, 1), draw(1, 2), draw(1, 2), draw(-1, 1)
    rows = np.stack([ndoth, alpha, ldoth, ext, inn, ddotn], axis=1).copy()
    out = dev.debug_eval(DBG["BSDF_TERMS"], rows, 3)
    a = np.zeros(n, np.float32)
    lib.orc_beckmann(n, P(ndoth), P(alpha), P(a))
    near_or_equal(out[:, 0], a, 4e-6)
    lib.orc_fresnel(n, P(ldoth), P(ext), P(inn), P(a))
    nan = np.isnan(a)
    assert np.array_equal(np.isnan(out[:, 1]), nan) and np.array_equal(bits(out[:, 1])[~nan], bits(a)[~nan])  # sqrt + arithmetic
    for tex in (0, 1):
        uv = np.stack([draw(-3, 3), draw(-3, 3)], axis=1).copy()
        rows = np.zeros((n, 3), np.float32)
        rows[:, 0], rows[:, 1:3] = tex, uv
        out = dev.debug_eval(DBG["TEXTURE"], rows, 3)
        want = np.zeros((n, 3), np.float32)
        lib.orc_texture_value(flat.desc, tex, n, P(uv), P(want))
        assert np.array_equal(bits(out), bits(want)), tex
    dev.close()

    sc = scenes.small_bunny(32, 32, env_pixels=scenes.synthetic_env())
    flat = sc.initKdTree().flatten(1)
    dev = DeviceScene(flat)
    # directions: unit vectors, a quarter of them replaced by the poles, a unit in the last place beyond
    # the poles (acos gives NaN: row 0 on both sides, env_coord), the other axes, vectors with a NaN
    # component (column and row 0 on both sides) and the zero vector
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d = np.ascontiguousarray((d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32))
    special = np.array([[0, 1, 0], [0, -1, 0], [0, 1.0000001, 0], [0, -1.0000001, 0], [3e-4, 1.0000001, -2e-4], [1, 0, 0], [0, 0, 1], [0, 0, -1],
                        [np.nan, 0.5, 0.5], [0.5, np.nan, 0.5], [0.5, 0.5, np.nan], [0, 0, 0]], np.float32)
    m = rng.random(n) < 0.25
    d[m] = special[rng.integers(0, len(special), int(m.sum()))]
    g = dev.debug_eval(4, d, 7)  # IBL
    o = np.zeros((n, 7), np.float32)
    lib.orc_ibl(flat.desc, n, P(d), P(o))
    # texel lookups are discrete: identical except where atan2f / acosf rounding crosses a texel edge
    assert (np.abs(g[:, 0:6] - o[:, 0:6]).max(axis=1) == 0).mean() > 0.995
    beyond = (np.abs(d[:, 1]) > 1) | np.isnan(d).any(axis=1)
    assert beyond.sum() > 500 and np.array_equal(bits(g[beyond, 0:6]), bits(o[beyond, 0:6]))
    same_class(g[:, 6], o[:, 6])
    dev.close()
