"""Pins the CPU oracle against golden vectors recorded from the reference's own translation units
(tests/golden/make_golden.py) and against the reference's src/expected.h known-answer tables.

Arithmetic-only results (intersection, tree build, transforms, PCG32, stratified streams) must be
bit-exact.  Results that go through libm (exp, pow, sin, cos, acos, atan2) are compared to
2 ulp-ish (rel 5e-7) so that a different glibc does not fail the suite.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

from raytracer_amd import abi, scenes
from tests import orc
from tests.golden import cases

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
P = orc._p


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def gold_f(z, key):
    return z[key].view(np.float32)


@pytest.fixture(scope="module")
def fn():
    return np.load(os.path.join(GOLD, "functions.npz")), cases.function_inputs()


def close(a, b, rel=5e-7, abs_=1e-12):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    both_nan = np.isnan(a) & np.isnan(b)
    ok = both_nan | (np.abs(a - b) <= abs_ + rel * np.abs(b)) | (a == b)
    assert ok.all(), "max rel err %g at %d entries" % (np.nanmax(np.abs(a - b) / (np.abs(b) + 1e-30)), int((~ok).sum()))


# ------------------------------------------------------------------------------------------ expected.h
def test_expected_h_known_answers(built):
    kat = json.load(open(os.path.join(GOLD, "expected_h.json")))
    lib = orc.oracle()
    b = np.asarray(kat["beckmann_ndoth_alpha_res"], np.float32)
    out = np.zeros(len(b), np.float32)
    lib.orc_beckmann(len(b), P(np.ascontiguousarray(b[:, 0])), P(np.ascontiguousarray(b[:, 1])), P(out))
    # the table was printed with %f: 6 decimals
    assert np.max(np.abs(out - b[:, 2])) <= 5e-6
    f = np.asarray(kat["fresnel_ldoth_ext_int_res"], np.float32)
    out = np.zeros(len(f), np.float32)
    lib.orc_fresnel(len(f), P(np.ascontiguousarray(f[:, 0])), P(np.ascontiguousarray(f[:, 1])), P(np.ascontiguousarray(f[:, 2])), P(out))
    assert len(b) == 26 and len(f) == 81
    assert np.max(np.abs(out - f[:, 3])) <= 5e-6


# ------------------------------------------------------------------------------------------ transforms
def test_transform_matches_reference(built):
    gold = json.load(open(os.path.join(GOLD, "transforms.json")))
    for name, g in gold.items():
        ops = [tuple(o) for o in g["ops"]]
        m, inv, t = orc.oracle_transform(ops)
        assert bits(m).tolist() == g["m"], name
        assert bits(inv).tolist() == g["inv"], name
        assert bits(t).tolist() == g["t"], name


# ------------------------------------------------------------------------------------------ kd-tree + hits
@pytest.mark.parametrize("name", cases.SCENES)
def test_tree_and_hits_match_reference(built, name):
    summary = json.load(open(os.path.join(GOLD, "trees.json")))[name]
    z = np.load(os.path.join(GOLD, "scene_%s.npz" % name))
    sc, w, h = cases.make_scene(name)
    tree = sc.initKdTree()
    flat = tree.flatten(0)
    # the oracle's literal restatement of initKdTree
    od = orc.oracle_tree(flat.desc)
    canon = orc.canonical_from_dump(od)
    assert len(od["leaf"]) == summary["n_nodes"] and len(od["refs"]) == summary["n_refs"] and od["depth"] == summary["depth"]
    assert orc.tree_digest(canon) == summary["digest"]
    assert bits(od["bmin"][0]).tolist() == summary["root_min"] and bits(od["bmax"][0]).tolist() == summary["root_max"]
    if "tree_axis" in z:
        assert np.array_equal(canon[0], z["tree_axis"]) and np.array_equal(canon[1], z["tree_split"])
        assert np.array_equal(canon[2], z["tree_counts"]) and np.array_equal(canon[3], z["tree_refs"])
    # intersectKdTree
    o, d, tmin, tmax = cases.rays_for(flat, w, h, name)
    assert cases.digest(o, d, tmin, tmax) == bytes(z["ray_digest"]).decode(), "ray inputs drifted from the recorded ones"
    r = orc.oracle_intersect(flat.desc, o, d, tmin, tmax, False)
    hit = z["closest_hit"] == 1
    assert np.array_equal(r["hit"], z["closest_hit"])
    assert np.array_equal(r["prim"][hit], z["closest_prim"][hit])
    assert np.array_equal(bits(r["t"])[hit], z["closest_t"][hit])
    assert np.array_equal(bits(r["u"])[hit], z["closest_u"][hit]) and np.array_equal(bits(r["v"])[hit], z["closest_v"][hit])
    assert np.array_equal(r["outside"][hit], z["closest_outside"][hit])
    if name in ("veach", "transformed"):  # sphere uv / normals go through atan2/acos/sqrt only for uv; normals are arithmetic
        close(r["nrm"][hit], z["closest_nrm"].view(np.float32)[hit])
    else:
        assert np.array_equal(bits(r["nrm"])[hit], z["closest_nrm"][hit])
    s = orc.oracle_intersect(flat.desc, o, d, tmin, tmax, True)
    ahit = z["any_hit"] == 1
    assert np.array_equal(s["hit"], z["any_hit"])
    assert np.array_equal(bits(s["t"])[ahit], z["any_t"][ahit])


# ------------------------------------------------------------------------------------------ functions
def test_bsdf_terms(built, fn):
    z, g = fn
    lib, n = orc.oracle(), g["n"]
    a = np.zeros(n, np.float32)
    lib.orc_beckmann(n, P(g["ndoth"]), P(g["alpha"]), P(a))
    close(a, gold_f(z, "beckmann"))
    lib.orc_fresnel(n, P(g["ldoth_signed"]), P(g["ext"]), P(g["int"]), P(a))
    assert np.array_equal(bits(a), z["fresnel"])  # sqrt and arithmetic only
    lib.orc_smith(n, P(g["ldoth"]), P(g["ldotn"]), P(g["vdoth"]), P(g["vdotn"]), P(g["alpha"]), P(a))
    assert np.array_equal(bits(a), z["smith"])
    b = np.zeros((n, 3), np.float32)
    lib.orc_bsdf(n, P(g["bsdf_in"]), P(b))
    close(b, gold_f(z, "bsdf"))
    f0, df, F = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
    lib.orc_schlick_f(n, P(g["base"]), P(g["metal"]), P(g["ldoth"]), P(f0), P(df), P(F))
    assert np.array_equal(bits(f0), z["schlick_f0"]) and np.array_equal(bits(df), z["schlick_diffuse"])
    close(F, gold_f(z, "schlick_F"))


def test_warps_and_distributions(built, fn):
    z, g = fn
    lib, n = orc.oracle(), g["n"]
    a = np.zeros((n, 2), np.float32)
    lib.orc_uniform_sample_triangle(n, P(g["u2"]), P(a))
    assert np.array_equal(bits(a), z["uniform_sample_triangle"])
    lib.orc_concentric_disk(n, P(g["u2"]), P(a))
    close(a, gold_f(z, "concentric_disk"))
    b = np.zeros(n, np.float32)
    lib.orc_power_heuristic(n, P(g["pdf_f"]), P(g["pdf_g"]), P(b))
    assert np.array_equal(bits(b), z["power_heuristic"])
    for nl in cases.LIGHT_COUNTS:
        func = np.ones(nl, np.float32)
        off, pdf = np.zeros(n, np.int32), np.zeros(n, np.float32)
        lib.orc_dist1d_discrete(P(func), nl, n, P(g["u1"]), P(off), P(pdf))
        assert np.array_equal(off, z["dist1d_off_%d" % nl]) and np.array_equal(bits(pdf), z["dist1d_pdf_%d" % nl])
    env = scenes.synthetic_env(*cases.ENV_SIZE)
    lib.orc_dist2d_pdf(P(env.reshape(-1)), cases.ENV_SIZE[0], cases.ENV_SIZE[1], n, P(g["u2"]), P(b))
    assert np.array_equal(bits(b), z["dist2d_pdf"])


def test_rng_and_stratified_sampler(built, fn):
    z, g = fn
    lib = orc.oracle()
    a = np.zeros(256, np.uint32)
    lib.orc_rng_u32(cases.RNG_SEQ, 256, P(a))
    assert np.array_equal(a, z["rng_u32"])
    f = np.zeros(256, np.float32)
    lib.orc_rng_float(cases.RNG_SEQ, 256, P(f))
    assert np.array_equal(bits(f), z["rng_float"])
    lib.orc_rng_bounded(cases.RNG_SEQ, 256, P(g["bounds"]), P(a))
    assert np.array_equal(a, z["rng_bounded"])
    for (nx, ny, seed, npix, nget) in cases.STRATIFIED_CASES:
        out = np.zeros(npix * nx * ny * (5 + 7 * nget), np.float32)
        lib.orc_stratified_stream(nx, ny, 8, seed, npix, nget, P(out))
        assert np.array_equal(bits(out), z["stratified_%dx%d_%d" % (nx, ny, seed)]), "StratifiedSampler stream differs"


def test_camera(built, fn):
    z, g = fn
    lib, n = orc.oracle(), g["n"]
    for name, c in cases.CAMERA_CASES.items():
        frm, at, up = [np.asarray(x, np.float32) for x in c[:3]]
        cam = abi.Camera()
        lib.orc_camera(P(frm), P(at), P(up), *[float(x) for x in c[3:]], C.byref(cam))
        par = np.concatenate([np.array(cam.near_top_left, np.float32), np.array(cam.dx_pixel, np.float32),
                              np.array(cam.dy_pixel, np.float32), np.array(cam.x_dir, np.float32)])
        close(par, gold_f(z, "camera_%s_params" % name), rel=1e-6)  # tan()
        # rays are compared on the reference's own camera parameters so that only get_ray is under test
        gp = gold_f(z, "camera_%s_params" % name)
        cam.near_top_left[:], cam.dx_pixel[:], cam.dy_pixel[:], cam.x_dir[:] = gp[0:3], gp[3:6], gp[6:9], gp[9:12]
        o, d = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
        lib.orc_camera_rays(C.byref(cam), n, P(g["st"]), P(g["u2"]), P(o), P(d))
        close(o, gold_f(z, "camera_%s_o" % name), rel=2e-6, abs_=1e-7)
        close(d, gold_f(z, "camera_%s_d" % name), rel=2e-6, abs_=1e-7)


def test_aabb_and_single_shapes(built, fn):
    z, g = fn
    lib, n = orc.oracle(), g["n"]
    hit, t0, t1 = np.zeros(n, np.uint8), np.zeros(n, np.float32), np.zeros(n, np.float32)
    lib.orc_intersect_aabb(n, P(g["ray_o"]), P(g["ray_d"]), P(g["tmin"]), P(g["tmax"]), P(g["box_min"]), P(g["box_max"]), P(hit), P(t0), P(t1))
    assert np.array_equal(hit, z["aabb_hit"])
    h = hit == 1
    assert np.array_equal(bits(t0)[h], z["aabb_tmin"][h]) and np.array_equal(bits(t1)[h], z["aabb_tmax"][h])

    sc = cases.shape_scene()
    flat = sc.initKdTree().flatten(0)
    for obj, name in enumerate(cases.SHAPE_NAMES):
        r = orc.trace_out(n)
        lib.orc_object_intersect(flat.desc, obj, n, P(g["ray_o"]), P(g["ray_d"]), P(g["tmin"]), P(g["tmax"]), 0, P(r["hit"]), P(r["t"]),
                                 P(r["u"]), P(r["v"]), P(r["pos"]), P(r["nrm"]), P(r["outside"]))
        assert np.array_equal(r["hit"], z["shape_%s_hit" % name]), name
        h = r["hit"] == 1
        assert h.sum() > 50, name
        assert np.array_equal(bits(r["t"])[h], z["shape_%s_t" % name][h]), name
        assert np.array_equal(bits(r["pos"])[h], z["shape_%s_pos" % name][h]), name
        assert np.array_equal(bits(r["nrm"])[h], z["shape_%s_nrm" % name][h]), name
        assert np.array_equal(r["outside"][h], z["shape_%s_outside" % name][h]), name
        if name == "sphere":  # spherical uv: atan2 / acos
            close(r["u"][h], gold_f(z, "shape_sphere_u")[h], rel=2e-6, abs_=1e-6)
            close(r["v"][h], gold_f(z, "shape_sphere_v")[h], rel=2e-6, abs_=1e-6)
        else:
            assert np.array_equal(bits(r["u"])[h], z["shape_%s_u" % name][h]) and np.array_equal(bits(r["v"])[h], z["shape_%s_v" % name][h])


def test_light_shape_sampling(built, fn):
    """Object::sample / Area / pdf for a triangle and a sphere emitter."""
    z, g = fn
    lib, n = orc.oracle(), g["n"]
    sc = cases.shape_scene()
    # the same shapes as lights (a ShapeLight owns a private copy of its shape)
    src_flat = sc.initKdTree().flatten(0)  # keeps the arrays behind the description alive
    src = src_flat.desc.contents
    p = np.ctypeslib.as_array(src.tri_p, shape=(3, 9))[0].copy()
    nn = np.ctypeslib.as_array(src.tri_n, shape=(3, 9))[0].copy()
    sc.add_triangle_light(p, nn, (0, 0, 1), sc.transforms[0], 0, (1, 1, 1))
    sc.add_sphere_light(sc.transforms[1], 0, (1, 1, 1))
    flat = sc.initKdTree().flatten(0)
    for light, name in enumerate(("triangle", "sphere")):
        pos, nrm, pdf = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32), np.zeros(n, np.float32)
        lib.orc_light_shape_sample(flat.desc, light, n, P(g["surf_pos"]), P(g["surf_nrm"]), P(g["outside"]), P(g["u2"]), P(pos), P(nrm), P(pdf))
        close(pos, gold_f(z, "shape_%s_sample_pos" % name), rel=2e-6, abs_=1e-6)
        close(nrm, gold_f(z, "shape_%s_sample_nrm" % name), rel=2e-6, abs_=1e-6)
        close(pdf, gold_f(z, "shape_%s_sample_pdf" % name), rel=1e-6)
        a = np.zeros(n, np.float32)
        lib.orc_light_shape_pdf(flat.desc, light, n, P(g["surf_pos"]), P(g["surf_nrm"]), P(g["toward_%s" % name]), P(a))
        gp = gold_f(z, "shape_%s_pdf" % name)
        assert (gp > 0).sum() > 100
        assert np.array_equal(bits(a), z["shape_%s_pdf" % name]), name
        area = lib.orc_light_shape_area(flat.desc, light)
        assert np.float32(area).view(np.uint32) == z["shape_%s_area" % name][0]


# ------------------------------------------------------------------------------------------ live, container only
@pytest.mark.ref
def test_live_reference_large_mesh(built):
    """In the build container: a 69k-triangle tree and 100k rays against the reference itself."""
    sc = scenes.bunny_ibl(64, 36, env=None)
    tree = sc.initKdTree()
    flat = tree.flatten(1)
    rs = orc.RefScene(flat.desc, orc.xf_ops_for(flat, sc))
    rd = rs.dump()
    assert orc.tree_digest(orc.canonical_from_dump(rd)) == orc.tree_digest(orc.canonical_from_abi(*flat.nodes()))
    o, d = orc.random_rays(flat.desc, 100000, seed=77)
    n = len(o)
    tmin, tmax = np.zeros(n, np.float32), np.full(n, 1e5, np.float32)
    for shadow in (False, True):
        a = rs.intersect(o, d, tmin, tmax, shadow)
        b = orc.oracle_intersect(flat.desc, o, d, tmin, tmax, shadow)
        h = a["hit"] == 1
        assert np.array_equal(a["hit"], b["hit"])
        assert np.array_equal(bits(a["t"])[h], bits(b["t"])[h])
        if not shadow:
            assert np.array_equal(a["prim"][h], b["prim"][h])


# ------------------------------------------------------------------------------------------ shading layer,
# the pieces that build without Light.h: textures.hpp, random_dir, CoordinateSystem, LightDistribution.cpp
@pytest.fixture(scope="module")
def l2():
    return np.load(os.path.join(GOLD, "shading_l2.npz")), cases.shading_inputs()


def test_textures_match_reference(built, l2):
    """texture::value(u, v, duv) of a checker_texture and an image_texture (read back through the
    reference's loadPng) under a texture transform: arithmetic + table lookup, bit-exact."""
    z, g = l2
    lib = orc.oracle()
    flat = cases.texture_scene().initKdTree().flatten(0)
    n = g["n"]
    for tex, key in ((0, "checker_value"), (1, "image_value")):
        out = np.zeros((n, 3), np.float32)
        lib.orc_texture_value(flat.desc, tex, n, P(g["uv"]), P(out))
        assert np.array_equal(bits(out), z[key]), key
    # both colours of the checker and many texels are exercised
    assert len(np.unique(z["checker_value"][:, 0])) == 2 and len(np.unique(z["image_value"], axis=0)) > 200


def test_random_dir_matches_reference(built, l2):
    """random_dir fed with the engine draws recorded from the reference's own thread-local engine,
    in drawing order: pins the arithmetic and which draw feeds which component (g++ evaluates the
    vec3 constructor arguments last to first)."""
    z, g = l2
    n = g["n"]
    out = np.zeros((n, 3), np.float32)
    draws = np.ascontiguousarray(gold_f(z, "random_dir_draws"))
    orc.oracle().orc_random_dir_replay(n, P(g["normal"]), P(draws), P(out))
    close(out, gold_f(z, "random_dir"), rel=2e-6, abs_=1e-7)  # sin / cos / sqrt of glibc
    assert (bits(out) == z["random_dir"]).all(axis=1).mean() > 0.99
    # cosine-weighted about the normal
    assert (np.sum(out * g["normal"], axis=1) > -1e-6).all()


def test_coordinate_system_and_light_distribution_match_reference(built, l2):
    z, g = l2
    lib = orc.oracle()
    n = g["n"]
    a, b = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
    lib.orc_coordinate_system(n, P(g["normal"]), P(a), P(b))
    assert np.array_equal(bits(a), z["coordinate_system_v2"]) and np.array_equal(bits(b), z["coordinate_system_v3"])
    for nl in cases.LIGHT_COUNTS:
        off, pdf = np.zeros(n, np.int32), np.zeros(n, np.float32)
        lib.orc_uniform_light_distribution(nl, n, P(g["u1"]), P(off), P(pdf))
        assert np.array_equal(off, z["uniform_light_off_%d" % nl]) and np.array_equal(bits(pdf), z["uniform_light_pdf_%d" % nl])
        assert off.min() == 0 and off.max() == nl - 1


@pytest.mark.ref
@pytest.mark.parametrize("name", cases.SCENES)
def test_reference_side_adapter_flattens_like_the_product(built, name):
    """integration/flatten_ref.cpp (the adapter a maintainer drops into the reference tree) run on
    the reference's own Scene + KdTree gives, table for table, the description the product's
    flatten() makes of the same scene: objects, transforms, kd-tree nodes, references, root box,
    out-of-tree list, camera.  Container only: the adapter is compiled against the reference's headers."""
    lib = orc.ref()
    sc, w, h = cases.make_scene(name)
    tree = sc.initKdTree()
    flat = tree.flatten(0)  # pre-order with adjacent children: the adapter's node order
    d = flat.desc.contents
    rs = orc.RefScene(flat.desc, orc.xf_ops_for(flat, sc))
    cam = {"cornell": cases.CAMERA_CASES["scene9"][:3] + (60.0, 64.0, 64.0, 0.001, 8.1)}.get(name)
    if cam:
        frm, at, up = [np.asarray(x, np.float32) for x in cam[:3]]
        lib.ref_scene_set_camera(rs.scene, lib.ref_camera_new(P(frm), P(at), P(up), *[float(x) for x in cam[3:]]))
    h_ = lib.ref_adapter_flatten(rs.scene, rs.tree)
    assert h_, "the adapter refused the scene"
    a = lib.ref_adapter_desc(h_).contents
    assert (a.struct_size, a.n_objects, a.n_nodes, a.n_refs, a.n_out_of_tree, a.n_transforms) == \
           (d.struct_size, d.n_objects, d.n_nodes, d.n_refs, d.n_out_of_tree, d.n_transforms)
    n = d.n_objects

    def same(pa, pb, shape, dtype):
        x = np.ctypeslib.as_array(C.cast(pa, C.POINTER(dtype)), shape=shape)
        y = np.ctypeslib.as_array(C.cast(pb, C.POINTER(dtype)), shape=shape)
        return np.array_equal(x, y)

    assert same(a.obj_kind, d.obj_kind, (n,), C.c_uint8)
    assert same(a.obj_transform, d.obj_transform, (n,), C.c_uint32)
    for field, width in (("tri_p", 9), ("tri_n", 9), ("tri_uv", 6), ("sphere", 4)):
        assert same(getattr(a, field), getattr(d, field), (n * width,), C.c_uint32), field
    assert same(a.transforms, d.transforms, (d.n_transforms * 21,), C.c_uint32)
    assert same(a.nodes, d.nodes, (d.n_nodes * 2,), C.c_uint32)
    assert same(a.refs, d.refs, (d.n_refs,), C.c_uint32)
    if d.n_out_of_tree:
        assert same(a.out_of_tree, d.out_of_tree, (d.n_out_of_tree,), C.c_uint32)
    assert bits(np.array(a.root_min)).tolist() == bits(np.array(d.root_min)).tolist()
    assert bits(np.array(a.root_max)).tolist() == bits(np.array(d.root_max)).tolist()
    # RefScene gives every object its own material (index = object index) with the product's emission
    om = np.ctypeslib.as_array(a.obj_material, shape=(n,))
    assert np.array_equal(om, np.arange(n))
    pm = np.ctypeslib.as_array(d.obj_material, shape=(n,))
    for i in range(0, n, max(1, n // 50)):
        assert tuple(a.materials[i].emission) == tuple(d.materials[int(pm[i])].emission)
    if cam:
        assert a.has_camera == 1
        for f in ("pos", "near_top_left", "dx_pixel", "dy_pixel", "x_dir", "up"):
            close(np.array(getattr(a.camera, f), np.float32), np.array(getattr(d.camera, f), np.float32), rel=2e-6, abs_=1e-7)
    lib.ref_adapter_free(h_)


@pytest.mark.ref
def test_reference_side_adapter_flattens_textures(built):
    """The adapter on reference materials that carry a checker_texture and an image_texture of a PNG
    file (what scene 18 attaches, example_scenes.cpp:1244; the reference keeps the pixels in a private
    member): kind, size, colours, texture transform and every texel equal the product's tables for the
    same scene."""
    from raytracer_amd.host import Scene, Transform
    lib = orc.ref()
    png = scenes.asset("pf-s125-ake7011-a.png")
    ops_img = [("scale", 0.5, 2.0, 1.0), ("rotate", 0, 0, 1, 30.0)]
    ops_chk = [("scale", 0.25, 0.4, 1.0)]
    sc = Scene()
    sc.setSimpleCamera((0, 1, 6), (0, 0.5, 0), (0, 1, 0), 50, 64, 64, 0.0, 6.0)
    mats = [sc.CreateMaterial(abi.MRT_MAT_SPECULAR, albedo=(0, 0, 0), roughness=0.1, metalness=0.01), sc.CreateMaterial(), sc.CreateMaterial()]
    sc.set_image_texture_file(mats[0], png, cases.transform_from_ops(ops_img))
    sc.set_checker(mats[2], (0.7, 0, 0), (0.3, 0, 0), cases.transform_from_ops(ops_chk))
    xfs = [Transform().scale(0.5, 0.5, 0.5).translate((x, 0.5, 0)) for x in (-1.2, 0.0, 1.2)]
    for m, t in zip(mats, xfs):
        sc.add_sphere(t, m)
    flat = sc.initKdTree().flatten(0)
    d = flat.desc.contents
    rs = orc.RefScene(flat.desc, orc.xf_ops_for(flat, sc))
    # RefScene gives object i the material i
    hx = orc.ref_transform_handle(ops_img)
    lib.ref_material_set_image(rs.scene, 0, png.encode(), hx)
    hc = orc.ref_transform_handle(ops_chk)
    even, odd = np.array((0.7, 0, 0), np.float32), np.array((0.3, 0, 0), np.float32)
    lib.ref_material_set_checker(rs.scene, 2, P(even), P(odd), hc)
    h_ = lib.ref_adapter_flatten(rs.scene, rs.tree)
    assert h_, "the adapter refused the scene"
    a = lib.ref_adapter_desc(h_).contents
    assert (a.n_textures, a.n_texels) == (d.n_textures, d.n_texels) == (2, 3 * 800 * 533)
    assert [a.materials[i].texture for i in range(3)] == [d.materials[i].texture for i in range(3)] == [0, -1, 1]
    for i in range(2):
        ta, td = a.textures[i], d.textures[i]
        assert (ta.kind, ta.width, ta.height, ta.texel_offset) == (td.kind, td.width, td.height, td.texel_offset)
        for f in ("even", "odd"):
            assert bits(np.array(getattr(ta, f))).tolist() == bits(np.array(getattr(td, f))).tolist()
        for f in ("m", "inv", "t"):
            assert bits(np.array(getattr(ta.xf, f))).tolist() == bits(np.array(getattr(td.xf, f))).tolist(), (i, f)
    xa = np.ctypeslib.as_array(C.cast(a.texels, abi.u32p), shape=(a.n_texels,))
    xd = np.ctypeslib.as_array(C.cast(d.texels, abi.u32p), shape=(d.n_texels,))
    assert np.array_equal(xa, xd)
    lib.ref_adapter_free(h_)


@pytest.mark.ref
@pytest.mark.parametrize("seed", list(range(8)))
def test_live_reference_random_scenes_with_ties(built, seed):
    """Container only: on the seeded random scenes of tests/fuzz_scenes.py (duplicates, coplanar
    overlaps, grid-aligned vertices, slivers, emissive primitives, transformed spheres, planes) the
    product's tree equals the tree the reference's initKdTree builds from the same objects, and the
    oracle's hits equal the reference's intersectKdTree bit for bit, closest and shadow rays."""
    from tests import fuzz_scenes
    sc = fuzz_scenes.make(seed)
    tree = sc.initKdTree(1 + seed % 3)
    flat = tree.flatten(0)
    rs = orc.RefScene(flat.desc, orc.xf_ops_for(flat, sc))
    assert orc.tree_digest(orc.canonical_from_dump(rs.dump())) == orc.tree_digest(orc.canonical_from_abi(*flat.nodes()))
    o, d, tmin, tmax = fuzz_scenes.rays(flat, seed)
    o2, d2, tmin2, tmax2 = fuzz_scenes.rays_hard(flat, seed)  # from vertices, along edges, -0.0 components, empty intervals
    o, d, tmin, tmax = np.concatenate([o, o2]), np.concatenate([d, d2]), np.concatenate([tmin, tmin2]), np.concatenate([tmax, tmax2])
    for shadow in (False, True):
        a = rs.intersect(o, d, tmin, tmax, shadow)
        b = orc.oracle_intersect(flat.desc, o, d, tmin, tmax, shadow)
        h = a["hit"] == 1
        assert np.array_equal(a["hit"], b["hit"])
        assert np.array_equal(bits(a["t"])[h], bits(b["t"])[h])
        if not shadow:
            assert np.array_equal(a["prim"][h], b["prim"][h])
            assert np.array_equal(bits(a["u"])[h], bits(b["u"])[h]) and np.array_equal(bits(a["v"])[h], bits(b["v"])[h])
            assert np.array_equal(a["outside"][h], b["outside"][h])


@pytest.mark.ref
def test_live_reference_functions_on_edge_values(built):
    """Container only: the oracle's bsdf.cpp / sampling restatements against the reference's own
    functions on zero, -0, denormal, tiny, huge, infinite and NaN arguments (a quarter of every column):
    bit for bit, a NaN where the reference gives a NaN."""
    o, r = orc.oracle(), orc.ref()
    rng = np.random.default_rng(77)
    n = 100000
    edge = np.array([0, -0.0, 1, -1, 0.5, 1e-8, 1e-20, 1e-38, 1e-45, 0.002, 4e-6, 0.9999999, 1.0000001, 2, 1e8, 1e30, np.inf, -np.inf, np.nan],
                    np.float32)

    def draw(lo, hi):
        a = rng.uniform(lo, hi, n).astype(np.float32)
        m = rng.random(n) < 0.25
        a[m] = edge[rng.integers(0, len(edge), int(m.sum()))]
        return a

    def same(x, y):
        return bool(((bits(x) == bits(y)) | (np.isnan(x) & np.isnan(y))).all())

    a, b = np.zeros(n, np.float32), np.zeros(n, np.float32)
    ndoth, alpha = draw(-1, 1), draw(0, 1)
    o.orc_beckmann(n, P(ndoth), P(alpha), P(a)), r.ref_beckmann(n, P(ndoth), P(alpha), P(b))
    assert same(a, b)
    l, e, i = draw(-1, 1), draw(1, 2), draw(1, 2)
    o.orc_fresnel(n, P(l), P(e), P(i), P(a)), r.ref_fresnel(n, P(l), P(e), P(i), P(b))
    assert same(a, b)
    x = [draw(-1, 1) for _ in range(4)] + [draw(0, 1)]
    o.orc_smith(n, *[P(v) for v in x], P(a)), r.ref_smith(n, *[P(v) for v in x], P(b))
    assert same(a, b)
    rows = np.stack([draw(-1, 1) for _ in range(12)], axis=1).copy()
    A, B = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
    o.orc_bsdf(n, P(rows), P(A)), r.ref_bsdf(n, P(rows), P(B))
    assert same(A, B)
    base, metal, ld = np.stack([draw(0, 1) for _ in range(3)], axis=1).copy(), draw(0, 1), draw(-1, 1)
    oo, rr = [np.zeros((n, 3), np.float32) for _ in range(3)], [np.zeros((n, 3), np.float32) for _ in range(3)]
    o.orc_schlick_f(n, P(base), P(metal), P(ld), *[P(v) for v in oo]), r.ref_schlick_f(n, P(base), P(metal), P(ld), *[P(v) for v in rr])
    assert all(same(p, q) for p, q in zip(oo, rr))
    u2 = np.stack([draw(0, 1), draw(0, 1)], axis=1).copy()
    A2, B2 = np.zeros((n, 2), np.float32), np.zeros((n, 2), np.float32)
    o.orc_uniform_sample_triangle(n, P(u2), P(A2)), r.ref_uniform_sample_triangle(n, P(u2), P(B2))
    assert same(A2, B2)
    o.orc_concentric_disk(n, P(u2), P(A2)), r.ref_concentric_disk(n, P(u2), P(B2))
    assert same(A2, B2)
    f, g = draw(0, 10), draw(0, 10)
    o.orc_power_heuristic(n, P(f), P(g), P(a)), r.ref_power_heuristic(n, P(f), P(g), P(b))
    assert same(a, b)


@pytest.mark.ref
def test_live_reference_random_cameras(built):
    """Container only: 200 random SimpleCamera set-ups (position, target, up vectors that are not
    unit or not perpendicular, field of view, frame sizes down to one pixel, lens and focus): the
    parameters the host library and the oracle derive are the reference's bit for bit, and so are the
    rays the oracle's get_ray makes from them for random film and lens samples."""
    from raytracer_amd.host import Scene, Transform
    lib, o = orc.ref(), orc.oracle()
    rng = np.random.default_rng(1)
    fields = ("near_top_left", "dx_pixel", "dy_pixel", "x_dir")
    n = 256
    for k in range(200):
        frm, at = (rng.random(3) * 10 - 5).astype(np.float32), (rng.random(3) * 4 - 2).astype(np.float32)
        up = np.array([0, 1, 0], np.float32) if k % 2 else (rng.random(3) - 0.5).astype(np.float32)
        fov, lens, focus = float(np.float32(rng.uniform(10, 120))), float(np.float32(rng.uniform(0, 0.1))), float(np.float32(rng.uniform(0.5, 20)))
        w, h = int(rng.integers(1, 2000)), int(rng.integers(1, 1200))
        cam = lib.ref_camera_new(P(frm), P(at), P(up), fov, float(w), float(h), lens, focus)
        want = np.zeros(12, np.float32)
        lib.ref_camera_params(cam, P(want))
        sc = Scene()
        sc.add_sphere(Transform(), sc.CreateMaterial())
        sc.setSimpleCamera(tuple(map(float, frm)), tuple(map(float, at)), tuple(map(float, up)), fov, w, h, lens, focus)
        flat = sc.initKdTree(1).flatten(0)
        c = flat.desc.contents.camera
        assert np.array_equal(bits(np.concatenate([np.array(getattr(c, f), np.float32) for f in fields])), bits(want)), k
        oc = abi.Camera()
        o.orc_camera(P(frm), P(at), P(up), fov, float(w), float(h), lens, focus, C.byref(oc))
        assert np.array_equal(bits(np.concatenate([np.array(getattr(oc, f), np.float32) for f in fields])), bits(want)), k
        st = (rng.random((n, 2)) * np.array([w, h])).astype(np.float32)
        u2 = rng.random((n, 2)).astype(np.float32)
        ro, rd, oo, od = (np.zeros((n, 3), np.float32) for _ in range(4))
        lib.ref_camera_rays(cam, n, P(st), P(u2), P(ro), P(rd))
        o.orc_camera_rays(C.byref(oc), n, P(st), P(u2), P(oo), P(od))
        assert np.array_equal(bits(oo), bits(ro)) and np.array_equal(bits(od), bits(rd)), k


@pytest.mark.ref
def test_operand_order_of_the_bsdf_branch_cosine_as_gxx_builds_it(built):
    """Compiler evidence, not parity.  integrator.cpp:96-98 multiplies sample(..., &wi, ...) -- which
    writes wi -- by max(dot(normal, wi), 0) -- which reads it -- and C++ leaves the order of the two
    operands open.  oracle/ref/eval_order_*.cpp give the class of the reference's own Material.h an
    instrumented sample() in a translation unit of its own and evaluate a statement of exactly that
    shape with oracle/Makefile's flags: g++ takes the cosine first, i.e. from the light-sample
    direction still in wi.  That is the order the oracle (orc_shading.h) and the kernels
    (device_render.hpp: cos_prev_wi) reproduce; if a compiler ever answers 0 here, both follow the
    other order and this test says so."""
    import ctypes as C
    lib = C.CDLL(orc.REF_SO)
    assert lib.ref_eval_order_probe() == 1
