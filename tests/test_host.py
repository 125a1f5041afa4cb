"""CPU tests of the product's host library (libmrt_host.so) and of the C-ABI surface.

The host library is product code; the oracle and the golden vectors are the checkers:
  - Transform arithmetic against the reference's matrices (golden),
  - initKdTree (multi-threaded, restructured) against the reference tree digests (golden) and
    against the oracle's literal restatement, for every thread count,
  - flatten(): both node layouts describe the same tree; description fields are consistent,
  - the loader (OBJ/MTL/Radiance HDR) on the committed assets,
  - libmrt_hip.so loads without a GPU, exports every symbol of include/mrt.h and rejects
    malformed scene descriptions before touching the device.
"""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

from raytracer_amd import abi, scenes
from raytracer_amd.host import Scene, Transform
from tests import orc
from tests.golden import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_transform_matches_reference(built):
    gold = json.load(open(os.path.join(GOLD, "transforms.json")))
    for name, g in gold.items():
        t = Transform()
        for op in g["ops"]:
            if op[0] == "scale":
                t.scale(*op[1:4])
            elif op[0] == "rotate":
                t.rotate(op[1:4], op[4])
            else:
                t.translate(op[1:4])
        m, inv, tr = t.get()
        assert bits(m).tolist() == g["m"], name
        assert bits(inv).tolist() == g["inv"], name
        assert bits(tr).tolist() == g["t"], name


@pytest.mark.parametrize("name", cases.SCENES)
def test_kdtree_identical_to_reference(built, name):
    summary = json.load(open(os.path.join(GOLD, "trees.json")))[name]
    sc, w, h = cases.make_scene(name)
    digests = set()
    for threads in (1, 3, 8):
        tree = sc.initKdTree(threads)
        nn, nr, depth = tree.counts()
        assert (nn, nr, depth) == (summary["n_nodes"], summary["n_refs"], summary["depth"])
        digests.add(orc.tree_digest(orc.canonical_from_dump(tree.dump())))
        for layout in (0, 1):
            flat = tree.flatten(layout)
            assert flat.depth == depth
            digests.add(orc.tree_digest(orc.canonical_from_abi(*flat.nodes())))
            d = flat.desc.contents
            assert bits(np.array(d.root_min)).tolist() == summary["root_min"]
            assert bits(np.array(d.root_max)).tolist() == summary["root_max"]
    assert digests == {summary["digest"]}


def test_flatten_description_is_consistent(built):
    sc, w, h = cases.make_scene("transformed")
    flat = sc.initKdTree().flatten(1)
    d = flat.desc.contents
    assert d.struct_size == C.sizeof(abi.SceneDesc)
    assert d.n_objects == sc.n_objects and d.n_materials == sc.n_materials and d.n_lights == sc.n_lights == 1
    t0 = d.transforms[0]
    assert list(t0.m) == [1, 0, 0, 0, 1, 0, 0, 0, 1] and list(t0.t) == [0, 0, 0]
    kinds = np.ctypeslib.as_array(d.obj_kind, shape=(d.n_objects,))
    assert set(kinds.tolist()) == {abi.MRT_TRIANGLE, abi.MRT_SPHERE, abi.MRT_PLANE}
    out = np.ctypeslib.as_array(d.out_of_tree, shape=(d.n_out_of_tree,))
    assert (kinds[out] == abi.MRT_PLANE).all() and d.n_out_of_tree == 1
    nodes, refs = flat.nodes()
    leaves = (nodes[:, 1] & 3) == 3
    assert (kinds[refs] != abi.MRT_PLANE).all()
    assert int((nodes[leaves, 1] >> 2).sum()) == len(refs)
    inner = np.nonzero(~leaves)[0]
    assert ((nodes[inner, 1] >> 2) > inner).all(), "children are stored after their parent"
    light = d.lights[0]
    assert light.kind == abi.MRT_LIGHT_SHAPE and light.shape_kind == abi.MRT_SPHERE
    assert d.has_camera == 1 and d.env_light == -1


def test_sphere_geometry_matches_oracle(built):
    t = Transform().scale(0.3, 0.9, 0.5).rotate((0, 1, 0), 12).translate((1, 2, 3))
    sc = Scene()
    m = sc.CreateMaterial()
    sc.add_sphere(t, m)
    flat = sc.initKdTree().flatten(0)
    d = flat.desc.contents
    got = np.ctypeslib.as_array(d.sphere, shape=(1, 4))[0]
    want = np.zeros(4, np.float32)
    orc.oracle().orc_sphere_from_transform(C.byref(d.transforms[1]), orc._p(want))
    assert np.array_equal(bits(got), bits(want))
    assert abs(got[3] - 0.9) < 1e-6


def test_obj_mtl_loader_on_cornell_box(built):
    sc = scenes.cornell(32, 32)
    # 32 triangles, one Material per face plus the default one, two emissive faces -> two ShapeLights
    assert sc.n_objects == 32 and sc.n_materials == 33 and sc.n_lights == 2
    flat = sc.initKdTree().flatten(0)
    d = flat.desc.contents
    em = [tuple(d.materials[i].emission) for i in range(d.n_materials)]
    assert sum(1 for e in em if e != (0.0, 0.0, 0.0)) == 2
    for i in range(2):
        assert d.lights[i].kind == abi.MRT_LIGHT_SHAPE and d.lights[i].shape_kind == abi.MRT_TRIANGLE
        assert tuple(d.lights[i].color) == (1.0, 1.0, 1.0)
    # addObjectsFromFile sets roughness 1, metalness 0 for MTL materials
    assert d.materials[1].roughness == 1.0 and d.materials[1].metalness == 0.0


def test_obj_loader_prescale_and_subdivision(built):
    a = Scene()
    m = a.CreateMaterial()
    n0 = a.addObjectsFromFile(scenes.asset("veach_mi/floor.obj"), m, Transform())
    b = Scene()
    m = b.CreateMaterial()
    n2 = b.addObjectsFromFile(scenes.asset("veach_mi/floor.obj"), m, Transform(), prescale=2.0, subdivisions=2)
    assert n2 == n0 * 16
    fa, fb = a.initKdTree().flatten(0), b.initKdTree().flatten(0)
    pa = np.ctypeslib.as_array(fa.desc.contents.tri_p, shape=(n0, 9)).reshape(-1, 3)
    pb = np.ctypeslib.as_array(fb.desc.contents.tri_p, shape=(n2, 9)).reshape(-1, 3)
    assert np.allclose(pb.min(0), 2 * pa.min(0)) and np.allclose(pb.max(0), 2 * pa.max(0))


def test_hdr_loader(built):
    sc = Scene()
    sc.add_ibl_file(scenes.asset("brown_photostudio_06_1k.hdr"))
    m = sc.CreateMaterial()
    sc.add_sphere(Transform(), m)
    flat = sc.initKdTree().flatten(0)
    d = flat.desc.contents
    assert (d.env_width, d.env_height) == (1024, 512) and d.env_light == 0
    env = np.ctypeslib.as_array(d.env_rgb, shape=(512, 1024, 3))
    assert np.isfinite(env).all() and env.min() >= 0 and 0.01 < env.mean() < 100
    # RGBE decode: every value is mantissa * 2^(e-136), i.e. has at most 8 significant bits
    m, e = np.frexp(env[env > 0][:100000])
    assert np.all((m * 256) == np.round(m * 256))
    with pytest.raises(Exception):
        Scene().add_ibl_file(scenes.asset("cornell-box.obj"))


def test_plant_obj_mtl_one_material_per_triangle(built):
    """indoor_plant_02.obj: 22 920 quads with an MTL file -> 45 840 triangles, one Material per
    triangle (example_scenes.cpp:85-92), no lights; the colour map is stored once."""
    sc = scenes.plant(64, 36)
    assert sc.n_objects == 45840
    assert sc.n_materials == 45841 and sc.n_lights == 1  # the environment
    m = sc.material(7)
    assert tuple(round(x, 2) for x in m.albedo) == (0.64, 0.64, 0.64) and m.roughness == 1.0 and m.metalness == 0.0
    assert m.texture == 0
    flat = sc.initKdTree().flatten(1)
    d = flat.desc.contents
    assert d.n_textures == 1 and d.textures[0].width == 1024 and d.textures[0].height == 1024
    mats = np.ctypeslib.as_array(d.obj_material, (d.n_objects,))
    assert len(np.unique(mats)) == 45840


def test_missing_obj_reports_error(built):
    sc = Scene()
    m = sc.CreateMaterial()
    with pytest.raises(Exception):
        sc.addObjectsFromFile("/nonexistent/file.obj", m, Transform())


# ---------------------------------------------------------------------------------------- C-ABI surface
def test_hip_library_exports_every_declared_symbol(built):
    hdr = open(os.path.join(ROOT, "include", "mrt.h")).read()
    names = set(re.findall(r"^(?:mrt_status|void|const char\*|int32_t)\s+(mrt_[a-z_]+)\s*\(", hdr, re.M))
    assert {"mrt_scene_create", "mrt_scene_destroy", "mrt_intersect", "mrt_intersect_noalloc", "mrt_render", "mrt_resolve", "mrt_last_error",
            "mrt_version", "mrt_debug_eval"} <= names
    lib = abi.hip()
    for n in names:
        assert hasattr(lib, n), n
    assert lib.mrt_version() == 2
    hdr = open(os.path.join(ROOT, "include", "mrt_host_c.h")).read()
    hlib = abi.host()
    host_names = set(re.findall(r"^[a-z_0-9 \*]+?\b(mrth_[a-z_]+)\s*\(", hdr, re.M))
    assert len(host_names) > 30
    for n in host_names:
        assert hasattr(hlib, n), n


def test_scene_create_rejects_malformed_descriptions(built):
    """Validation happens on the host before any device call, so it is testable without a GPU."""
    lib = abi.hip()
    sc = scenes.cornell(16, 16)
    flat = sc.initKdTree().flatten(1)
    h = C.c_void_p()

    def try_create(mutate):
        d = abi.SceneDesc()
        C.memmove(C.byref(d), flat.desc, C.sizeof(abi.SceneDesc))
        keep = mutate(d)
        st = lib.mrt_scene_create(C.byref(d), 0, C.byref(h))
        return st, lib.mrt_last_error().decode(), keep

    st, msg, _ = try_create(lambda d: setattr(d, "struct_size", 12))
    assert st == 1 and "struct_size" in msg

    def bad_child(d):
        nodes = (abi.KdNode * d.n_nodes)()
        C.memmove(nodes, d.nodes, C.sizeof(nodes))
        nodes[0].w1 = (nodes[0].w1 & 3) | ((d.n_nodes + 5) << 2)
        d.nodes = nodes
        return nodes
    st, msg, _ = try_create(bad_child)
    assert st == 1 and "child" in msg

    def bad_ref(d):
        refs = (C.c_uint32 * d.n_refs)()
        C.memmove(refs, d.refs, C.sizeof(refs))
        refs[0] = d.n_objects + 7
        d.refs = refs
        return refs
    st, msg, _ = try_create(bad_ref)
    assert st == 1 and "out of range" in msg

    def bad_material(d):
        m = (C.c_uint32 * d.n_objects)()
        C.memmove(m, d.obj_material, C.sizeof(m))
        m[3] = d.n_materials
        d.obj_material = m
        return m
    def sphere_out_of_tree(d):
        # kdtree.cpp:64: only planes stay outside the tree
        kinds = (C.c_uint32 * d.n_objects)()
        C.memmove(kinds, d.obj_kind, C.sizeof(kinds))
        oot = (C.c_uint32 * 1)(0)
        d.out_of_tree, d.n_out_of_tree = oot, 1
        return oot
    st, msg, _ = try_create(sphere_out_of_tree)
    assert st == 1 and "planes only" in msg

    def shared_child(d):
        # an inner node whose children are its parent's children: every index is in range and the
        # children follow their parent, but the tree is not a tree
        nodes = (abi.KdNode * d.n_nodes)()
        C.memmove(nodes, d.nodes, C.sizeof(nodes))
        root_left = nodes[0].w1 >> 2
        inner = next(i for i in range(1, d.n_nodes) if (nodes[i].w1 & 3) != 3 and (nodes[i].w1 >> 2) > root_left + 1)
        other = next(i for i in range(inner + 1, d.n_nodes) if (nodes[i].w1 & 3) != 3)
        nodes[other].w1 = (nodes[other].w1 & 3) | ((nodes[inner].w1 >> 2) << 2)
        if (nodes[other].w1 >> 2) <= other:
            return None
        d.nodes = nodes
        return nodes
    st, msg, keep = try_create(shared_child)
    if keep is not None:
        assert st == 1 and ("two parents" in msg or "visits more nodes" in msg), (st, msg)

    def too_many_nodes(d):
        d.n_nodes = 1 << 29
    st, msg, _ = try_create(too_many_nodes)
    assert st == 4 and "too large" in msg

    st, msg, _ = try_create(bad_material)
    assert st in (1, 2)  # material check follows the device probe on a machine without a GPU
    assert lib.mrt_scene_create(None, 0, C.byref(h)) == 1


def _copy_array(d, field, ctype, count):
    arr = (ctype * max(count, 1))()
    src = getattr(d, field)
    if count and src:
        C.memmove(arr, src, C.sizeof(ctype) * count)
    setattr(d, field, C.cast(arr, type(src)))
    return arr


def test_scene_create_validates_everything_a_kernel_indexes_with(built):
    """Materials, textures, lights, the environment and object kinds are checked on the host, before
    any device call; then 400 seeded corruptions of the index-carrying tables of a valid description
    (tree nodes, leaf references, out-of-tree list, per-object indices, material / texture / light
    records): every call returns a status -- malformed, unsupported, or, being valid still, the
    missing device (success on a machine with a GPU) -- and none crashes."""
    from tests import fuzz_scenes
    lib = abi.hip()
    for seed in range(64):  # the first scene with an environment, a shape light and a texture
        sc = fuzz_scenes.make_shaded(seed)
        flat = sc.initKdTree(2).flatten(1)
        desc = flat.desc.contents
        if desc.n_textures >= 1 and desc.env_light >= 0 and any(desc.lights[i].kind == abi.MRT_LIGHT_SHAPE for i in range(desc.n_lights)):
            break
    else:
        raise AssertionError("no seed gives a scene with all three")
    h = C.c_void_p()

    def try_create(mutate):
        d = abi.SceneDesc()
        C.memmove(C.byref(d), flat.desc, C.sizeof(abi.SceneDesc))
        keep = mutate(d)
        st = lib.mrt_scene_create(C.byref(d), 0, C.byref(h))
        if st == 0:
            lib.mrt_scene_destroy(h)
        return st, lib.mrt_last_error().decode(), keep

    valid = (0, 2)  # created, or no device
    assert try_create(lambda d: None)[0] in valid

    def edit(field, ctype, count_field, index, attr, value):
        def mutate(d):
            arr = _copy_array(d, field, ctype, getattr(d, count_field))
            if attr is None:
                arr[index] = value
            else:
                setattr(arr[index], attr, value(d) if callable(value) else value)
            return arr
        return mutate

    shape_light = next(i for i in range(desc.n_lights) if desc.lights[i].kind == abi.MRT_LIGHT_SHAPE)
    textured = next(i for i in range(desc.n_materials) if desc.materials[i].texture >= 0)
    for mutate, status, text in [
        (edit("materials", abi.Material, "n_materials", textured, "texture", lambda d: d.n_textures), 1, "material texture"),
        (edit("materials", abi.Material, "n_materials", 0, "type", 3), 1, "material type"),
        (edit("textures", abi.Texture, "n_textures", 0, "kind", 7), 1, "texture kind"),
        (edit("lights", abi.Light, "n_lights", shape_light, "kind", 9), 1, "light kind"),
        (edit("lights", abi.Light, "n_lights", shape_light, "shape_kind", abi.MRT_PLANE), 1, "triangle or a sphere"),
        (edit("lights", abi.Light, "n_lights", shape_light, "transform", lambda d: d.n_transforms), 1, "light transform"),
        (lambda d: setattr(d, "env_light", d.n_lights), 1, "env_light out of range"),
        (lambda d: setattr(d, "env_light", shape_light), 1, "must name an IBL"),
        (lambda d: setattr(d, "env_width", 0), 1, "without environment pixels"),
        (lambda d: setattr(d, "env_width", 1 << 16), 4, "too large"),
        (edit("obj_kind", C.c_uint8, "n_objects", 5, None, 4), 4, "object kind"),  # CUBE (geometry.h:5)
        (edit("obj_transform", C.c_uint32, "n_objects", 5, None, 1 << 20), 1, "object transform"),
    ]:
        st, msg, _ = try_create(mutate)
        assert st == status and text in msg, (st, msg, text)

    def image_texture(width, height, offset):
        def mutate(d):
            arr = _copy_array(d, "textures", abi.Texture, d.n_textures)
            arr[0].kind, arr[0].width, arr[0].height, arr[0].texel_offset = abi.MRT_TEX_IMAGE, width, height, offset
            texels = (C.c_float * 48)()
            d.texels, d.n_texels = C.cast(texels, type(d.texels)), 48
            return arr, texels
        return mutate
    assert try_create(image_texture(4, 4, 0))[0] in valid
    for w, hgt, off in [(4, 4, 3), (5, 4, 0), (0, 4, 0), (4, 4, 2 ** 64 - 3), (65535, 65535, 0)]:
        st, msg, _ = try_create(image_texture(w, hgt, off))
        assert st == 1 and "texel pool" in msg, (w, hgt, off, st, msg)
    # 65536 and beyond: refused outright (3 * w * h of 2^31 x 2863311531 wraps 64 bits to a window that "fits")
    for w, hgt in [(65536, 65536), (2 ** 32 - 1, 2 ** 32 - 1), (2 ** 31, 2863311531), (65536, 1)]:
        st, msg, _ = try_create(image_texture(w, hgt, 0))
        assert st == 4 and "too large" in msg, (w, hgt, st, msg)

    rng = np.random.default_rng(5)
    tables = [("nodes", C.c_uint32, lambda d: 2 * d.n_nodes), ("refs", C.c_uint32, lambda d: d.n_refs),
              ("out_of_tree", C.c_uint32, lambda d: d.n_out_of_tree), ("obj_material", C.c_uint32, lambda d: d.n_objects),
              ("obj_transform", C.c_uint32, lambda d: d.n_objects), ("obj_kind", C.c_uint8, lambda d: d.n_objects),
              ("materials", C.c_uint32, lambda d: d.n_materials * C.sizeof(abi.Material) // 4),
              ("textures", C.c_uint32, lambda d: d.n_textures * C.sizeof(abi.Texture) // 4),
              ("lights", C.c_uint32, lambda d: d.n_lights * C.sizeof(abi.Light) // 4)]
    seen = set()
    for trial in range(400):
        field, ctype, words = tables[trial % len(tables)]

        def corrupt(d):
            n = words(d)
            arr = (ctype * max(n, 1))()
            src = getattr(d, field)
            C.memmove(arr, src, C.sizeof(ctype) * n)
            for _ in range(int(rng.integers(1, 4))):
                k = int(rng.integers(0, n))
                how = int(rng.integers(0, 3))
                bits = 8 * C.sizeof(ctype)
                arr[k] = [int(arr[k]) ^ (1 << int(rng.integers(0, bits))), int(rng.integers(0, 1 << bits)), (1 << bits) - 1][how]
            setattr(d, field, C.cast(arr, type(src)))
            return arr
        st, msg, _ = try_create(corrupt)
        assert st in (0, 1, 2, 4), (trial, field, st, msg)
        seen.add(st)
    assert 1 in seen and (0 in seen or 2 in seen)


def test_product_path_fails_loudly_without_a_gpu(built):
    """No CPU fallback: on a machine without a HIP device the scene cannot be created and the
    renderer raises; nothing is routed through the oracle."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from raytracer_amd.host import DeviceScene, MrtError
    sc = scenes.cornell(16, 16)
    flat = sc.initKdTree().flatten(1)
    h = C.c_void_p()
    st = abi.hip().mrt_scene_create(flat.desc, 0, C.byref(h))
    assert st == 2 and not h.value, (st, abi.hip().mrt_last_error())
    with pytest.raises(MrtError):
        DeviceScene(flat)
    with pytest.raises(MrtError):
        sc.hip_render(16, 16, 1, 1)
    with pytest.raises(MrtError):
        sc.hip_render_progressive(16, 16, 2, 2, 1)


def test_intersect_and_render_reject_null_arguments(built):
    lib = abi.hip()
    assert lib.mrt_intersect(None, 0, None, None, 0, 0, None, None) == 1
    assert lib.mrt_render(None, None, None, 0, None, None) == 1
    assert b"NULL" in lib.mrt_last_error()


# ---------------------------------------------------------------------------------------- ingestion pinned
@pytest.mark.parametrize("name", cases.OBJ_FILES)
def test_obj_loader_matches_tinyobj_streams(built, name):
    """addObjectsFromFile of the host library against the per-face streams recorded from the
    reference's tiny_obj_loader.h + the face walk of example_scenes.cpp:32-81: positions, vertex
    normals (from the file, or the face normal when it has none), texture coordinates, and the
    per-face materials made from the MTL entries -- bit for bit, quads fanned as tinyobj fans them
    (primitive ids depend on the order)."""
    summary = json.load(open(os.path.join(GOLD, "obj_streams.json")))[name]
    z = np.load(os.path.join(GOLD, "obj_streams.npz"))
    key = summary["key"]
    sc = Scene()
    default = sc.CreateMaterial()
    n = sc.addObjectsFromFile(scenes.asset(name), default, Transform())
    assert n == summary["faces"]
    sc.setSimpleCamera((0, 0, 5), (0, 0, 0), (0, 1, 0), 40, 32, 32, 0.0, 5.0)
    flat = sc.initKdTree().flatten(0)  # keeps the arrays behind the description alive
    d = flat.desc.contents
    assert d.n_objects == n
    got = {"pos": np.ctypeslib.as_array(d.tri_p, shape=(n, 9)), "vertex_normals": np.ctypeslib.as_array(d.tri_n, shape=(n, 9)),
           "tex": np.ctypeslib.as_array(d.tri_uv, shape=(n, 6))}
    for k, a in got.items():
        assert cases.digest(np.ascontiguousarray(a, np.float32)) == summary["digest"][k], k
        if n <= cases.OBJ_FULL_LIMIT:
            assert np.array_equal(bits(a).reshape(n, -1), z["%s_%s" % (key, k)])
    om = np.ctypeslib.as_array(d.obj_material, shape=(n,))
    mtl = z[key + "_mtl"].view(np.float32).reshape(-1, 10)
    assert len(mtl) == summary["materials"]
    if len(mtl) == 0:
        assert (om == default).all()  # no MTL: every face gets the default material
        return
    # one new Material per face, in face order, after the default one (example_scenes.cpp:83-92)
    assert np.array_equal(om, default + 1 + np.arange(n))
    if n <= cases.OBJ_FULL_LIMIT:
        mid = z[key + "_material_id"]
        for f in range(n):
            m, want = d.materials[int(om[f])], mtl[mid[f]]
            assert (m.ior, tuple(m.albedo), tuple(m.emission)) == (want[0], tuple(want[1:4]), tuple(want[4:7]))
            mirror = tuple(want[7:10]) == (1.0, 1.0, 1.0)  # example_scenes.cpp:112-116
            assert m.type == (abi.MRT_MAT_SPECULAR if mirror else abi.MRT_MAT_DIFFUSE)
            assert (m.roughness, m.metalness) == ((np.float32(0.002), 1.0) if mirror else (1.0, 0.0))
        assert d.n_lights == int((mtl[mid][:, 4:7] != 0).any(axis=1).sum())


@pytest.mark.parametrize("name", ("bunny_pair", "million_bunny"))
def test_benchmark_scene_trees_identical_to_reference(built, name):
    """The kd-trees of the two benchmark scenes (138 902 and 1 111 216 triangles) built by the host
    library against the digests recorded from the reference's initKdTree."""
    summary = json.load(open(os.path.join(GOLD, "trees_large.json")))[name]
    sc = getattr(scenes, name)(64, 36)
    assert sc.n_objects == summary["n_objects"]
    tree = sc.initKdTree()
    assert tree.counts() == (summary["n_nodes"], summary["n_refs"], summary["depth"])
    assert orc.tree_digest(orc.canonical_from_dump(tree.dump())) == summary["digest"]
    flat = tree.flatten(1)
    d = flat.desc.contents
    assert bits(np.array(d.root_min)).tolist() == summary["root_min"] and bits(np.array(d.root_max)).tolist() == summary["root_max"]


@pytest.mark.ref
def test_obj_loader_random_files_against_the_references_tinyobj(built, tmp_path):
    """300 seeded OBJ + MTL files (tests/fuzz_obj.py: number spellings incl. the ones tinyobj's grammar
    refuses, white space, \\n / \\r\\n / \\r line ends, polygons of 3-7 corners, relative, zero and signed
    indices, groups and objects, repeated material names, statements before the first newmtl, mtllib
    lines with several names / a missing file / two blanks, usemtl straight before g or o) through
    the reference's tiny_obj_loader.h, live, and through addObjectsFromFile: the same faces in the
    same order, positions / vertex normals / texture coordinates bit for bit, the same per-face
    materials."""
    import sys
    sys.path.insert(0, GOLD)
    import make_golden
    from tests import fuzz_obj
    dropped = refused = 0
    for seed in range(300):
        path = fuzz_obj.make_obj(seed, str(tmp_path), "f%d" % seed)
        ref = make_golden.ref_obj_streams(path)
        text = open(path, newline="").read().replace("\r\n", "\n").replace("\r", "\n")
        fans = sum(max(len(l.split()) - 3, 0) for l in text.split("\n") if l.strip().startswith("f"))
        refused += any(w in text for w in (" nan", " abc", " -.75"))
        sc = Scene()
        default = sc.CreateMaterial()
        n = sc.addObjectsFromFile(path, default, Transform())
        assert n == len(ref["pos"]), seed
        dropped += n < fans
        if n == 0:
            continue
        sc.setSimpleCamera((0, 0, 5), (0, 0, 0), (0, 1, 0), 40, 32, 32, 0.0, 5.0)
        flat = sc.initKdTree(1).flatten(0)
        d = flat.desc.contents
        for k, a in (("pos", np.ctypeslib.as_array(d.tri_p, shape=(n, 9))), ("vertex_normals", np.ctypeslib.as_array(d.tri_n, shape=(n, 9))),
                     ("tex", np.ctypeslib.as_array(d.tri_uv, shape=(n, 6)))):
            assert np.array_equal(bits(a).reshape(n, -1), bits(ref[k]).reshape(n, -1)), (seed, k)
        om = np.ctypeslib.as_array(d.obj_material, shape=(n,))
        if len(ref["mtl"]) == 0:
            assert (om == default).all(), seed
            continue
        assert np.array_equal(om, default + 1 + np.arange(n)), seed
        for i in range(n):
            if ref["material_id"][i] < 0:
                continue  # the reference indexes materials[-1] for such a face: undefined
            m, r = d.materials[int(om[i])], ref["mtl"][ref["material_id"][i]]
            got = np.array([m.ior] + list(m.albedo) + list(m.emission), np.float32)
            want = np.array(r[:7], np.float32)
            if not want[4:7].any():
                want[4:7] = 0  # m_emission is only assigned when a component is non-zero (example_scenes.cpp:93-97): -0 stays +0
            assert np.array_equal(bits(got), bits(want)), (seed, i)
            assert (m.type == abi.MRT_MAT_SPECULAR) == bool(r[7] == 1 and r[8] == 1 and r[9] == 1), (seed, i)
    assert dropped >= 5 and refused >= 5  # the corpus does exercise those two behaviours


def test_obj_loader_random_files_against_recorded_tinyobj_digests(built, tmp_path):
    """The first 100 files of tests/fuzz_obj.py against digests of the streams the reference's
    tiny_obj_loader.h gave for them (tests/golden/obj_fuzz.json, written by make_golden.py): the same
    check as the live test above for machines without the reference build."""
    from tests import fuzz_obj
    gold = json.load(open(os.path.join(GOLD, "obj_fuzz.json")))
    for seed in range(100):
        g = gold[str(seed)]
        sc = Scene()
        default = sc.CreateMaterial()
        n = sc.addObjectsFromFile(fuzz_obj.make_obj(seed, str(tmp_path), "f%d" % seed), default, Transform())
        assert n == g["faces"], seed
        if n == 0:
            continue
        sc.setSimpleCamera((0, 0, 5), (0, 0, 0), (0, 1, 0), 40, 32, 32, 0.0, 5.0)
        flat = sc.initKdTree(1).flatten(0)
        d = flat.desc.contents
        for k, a in (("pos", np.ctypeslib.as_array(d.tri_p, shape=(n, 9))), ("vertex_normals", np.ctypeslib.as_array(d.tri_n, shape=(n, 9))),
                     ("tex", np.ctypeslib.as_array(d.tri_uv, shape=(n, 6)))):
            assert cases.digest(np.ascontiguousarray(a, np.float32)) == g["digest"][k], (seed, k)
        assert d.n_materials == 1 + (n if g["materials"] else 0), seed


def test_host_library_under_address_and_ub_sanitizers(built):
    """The host library's sources and the host side of the HIP library compiled with
    -fsanitize=address,undefined together with tests/sanitize/host_sanitize.cpp (scene set-up,
    OBJ/MTL/HDR/PPM/PNG loaders, multi-threaded kd-tree build, both flatten layouts, the C-ABI's
    validation on valid, malformed and randomly damaged descriptions): no report, leaks included
    (but for the 128 bytes the OpenMP runtime itself keeps, tests/sanitize/lsan.supp)."""
    import shutil
    import subprocess
    if not shutil.which("hipcc") or not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"):
        pytest.skip("no ROCm clang")
    from raytracer_amd import build
    exe = build.build_sanitize()
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1",
               LSAN_OPTIONS="suppressions=" + os.path.join(ROOT, "tests", "sanitize", "lsan.supp") + ":print_suppressions=0")
    # the three placements of the leaves' reference lists mrt_scene_create can build (in the pool of tree
    # records, line-aligned: the default; the same unaligned; in an array of their own)
    for lists in (None, "1", "0"):
        if lists is not None:
            env["MRT_REFS_INLINE"] = lists
        r = subprocess.run([exe, os.path.join(ROOT, "assets")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=600)
        assert r.returncode == 0 and "host_sanitize ok" in r.stdout, r.stdout[-3000:]
        assert "ERROR: AddressSanitizer" not in r.stdout and "runtime error" not in r.stdout and "LeakSanitizer" not in r.stdout, r.stdout[-3000:]


def _png_decode(path):
    """A minimal PNG reader for the files savePNG writes (8-bit RGB, filter 0): zlib does the inflate."""
    import struct
    import zlib
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w, h = 8, b"", 0, 0
    while pos < len(data):
        n, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        crc, = struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])
        assert zlib.crc32(typ + body) & 0xffffffff == crc, typ
        if typ == b"IHDR":
            w, h, depth, colour, comp, flt, lace = struct.unpack(">IIBBBBB", body)
            assert (depth, colour, comp, flt, lace) == (8, 2, 0, 0, 0)
        elif typ == b"IDAT":
            idat += body
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + 3 * w)
    assert not raw[:, 0].any()
    return raw[:, 1:].reshape(h, w, 3)


def test_png_writer(built, tmp_path):
    """savePNG: the 8-bit conversion of the reference's saveImage (ivec3(255.f * c) clamped,
    image.cpp:135-141) in a PNG that a standard inflate and CRC check accept; sizes that need several
    stored deflate blocks included."""
    rng = np.random.default_rng(8)
    for w, h in ((5, 3), (257, 131), (640, 360)):
        img = (rng.random((h, w, 3), dtype=np.float32) * 1.4 - 0.2).astype(np.float32)  # some values outside 0..1
        img[0, 0] = (0.0, 1.0, 0.999999)
        path = str(tmp_path / ("t%d.png" % w))
        assert abi.host().mrth_save_png(path.encode(), img.ctypes.data_as(abi.f32p), w, h) == 0
        want = np.clip((np.float32(255.0) * img).astype(np.int32), 0, 255).astype(np.uint8)
        assert np.array_equal(_png_decode(path), want)
    assert abi.host().mrth_save_png(b"/nonexistent-dir/x.png", img.ctypes.data_as(abi.f32p), w, h) != 0


@pytest.mark.ref
def test_png_writer_against_the_references_lodepng(built, tmp_path):
    """Container only: the file savePNG writes, read back by the reference's loadPng (lodepng), has
    the pixels of the file the reference's own saveImage writes for the same floats."""
    lib = orc.ref()
    rng = np.random.default_rng(9)
    w, h = 321, 123
    img = np.ascontiguousarray((rng.random((h, w, 3), dtype=np.float32) * 1.4 - 0.2).astype(np.float32))
    ours = str(tmp_path / "ours.png")
    assert abi.host().mrth_save_png(ours.encode(), img.ctypes.data_as(abi.f32p), w, h) == 0
    lib.ref_save_image(str(tmp_path / "theirs").encode(), w, h, orc._p(img))
    out = []
    for path in (ours, str(tmp_path / "theirs.png")):
        ww, hh = C.c_int(), C.c_int()
        px = np.zeros((h, w, 3), np.float32)
        assert lib.ref_load_png(path.encode(), C.byref(ww), C.byref(hh), orc._p(px), px.size) == 0
        assert (ww.value, hh.value) == (w, h)
        out.append(px)
    assert np.array_equal(out[0], out[1])


def _png_encode(path, px, colour, depth, palette=None, interlace=False, filters=(0, 1, 2, 3, 4), level=6, idat_split=0):
    """Writes samples px[h][w][channels] (integers below 2^depth) as a PNG with the given colour
    type, cycling through `filters` by scanline."""
    import struct
    import zlib
    h, w, ch = px.shape

    def pack_rows(img):
        hh, ww, _ = img.shape
        if depth == 16:
            rows = img.astype(">u2").reshape(hh, -1).view(np.uint8)
        elif depth == 8:
            rows = img.astype(np.uint8).reshape(hh, -1)
        else:
            per = 8 // depth
            flat = img.reshape(hh, ww).astype(np.uint8)
            pad = (-ww) % per
            flat = np.pad(flat, ((0, 0), (0, pad)))
            rows = np.zeros((hh, flat.shape[1] // per), np.uint8)
            for k in range(per):
                rows |= flat[:, k::per] << (8 - depth * (k + 1))
        return rows

    def filtered(rows, bpp):
        out = bytearray()
        prev = np.zeros(rows.shape[1], np.int32)
        for y in range(rows.shape[0]):
            cur = rows[y].astype(np.int32)
            a = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]]) if rows.shape[1] > bpp else np.zeros_like(cur)
            c = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]]) if rows.shape[1] > bpp else np.zeros_like(cur)
            f = filters[y % len(filters)]
            if f == 0:
                line = cur
            elif f == 1:
                line = cur - a
            elif f == 2:
                line = cur - prev
            elif f == 3:
                line = cur - ((a + prev) >> 1)
            else:
                p = a + prev - c
                pa, pb, pc = abs(p - a), abs(p - prev), abs(p - c)
                pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, prev, c))
                line = cur - pred
            out.append(f)
            out += (line & 255).astype(np.uint8).tobytes()
            prev = cur
        return bytes(out)

    bpp = max(1, ch * depth // 8)
    if interlace:
        raw = b""
        for x0, y0, dx, dy in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
            sub = px[y0::dy, x0::dx]
            if sub.shape[0] and sub.shape[1]:
                raw += filtered(pack_rows(sub), bpp)
    else:
        raw = filtered(pack_rows(px), bpp)
    z = zlib.compress(raw, level)

    def chunk(typ, body):
        return struct.pack(">I", len(body)) + typ + body + struct.pack(">I", zlib.crc32(typ + body) & 0xffffffff)

    data = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, colour, 0, 0, 1 if interlace else 0))
    data += chunk(b"gAMA", struct.pack(">I", 45455))
    if palette is not None:
        data += chunk(b"PLTE", palette.astype(np.uint8).tobytes())
    if idat_split:
        for i in range(0, len(z), idat_split):
            data += chunk(b"IDAT", z[i:i + idat_split])
    else:
        data += chunk(b"IDAT", z)
    data += chunk(b"IEND", b"")
    open(path, "wb").write(data)


def _png_cases(tmp_path):
    """(path, expected r, g, b bytes [h][w][3]) for every colour type / bit depth, interlaced or not."""
    rng = np.random.default_rng(21)
    out = []
    for colour, ch, depths in ((0, 1, (1, 2, 4, 8, 16)), (2, 3, (8, 16)), (3, 1, (1, 2, 4, 8)), (4, 2, (8, 16)), (6, 4, (8, 16))):
        for depth in depths:
            for interlace in (False, True):
                for w, h in ((1, 1), (7, 5), (33, 17)):
                    px = rng.integers(0, 1 << depth, (h, w, ch))
                    palette = rng.integers(0, 256, (1 << depth, 3)) if colour == 3 else None
                    s8 = (px >> 8) if depth == 16 else px  # lodepng keeps the high byte of 16-bit samples
                    if colour == 3:
                        want = palette[px[:, :, 0]]
                    elif colour in (0, 4):
                        g = s8[:, :, 0] if depth >= 8 else (px[:, :, 0] * 255) // ((1 << depth) - 1)
                        want = np.stack([g, g, g], axis=-1)
                    else:
                        want = s8[:, :, :3]
                    path = str(tmp_path / ("c%d_d%d_i%d_%dx%d.png" % (colour, depth, interlace, w, h)))
                    _png_encode(path, px, colour, depth, palette, interlace, level=(0 if w == 7 else 9 if w == 1 else 6), idat_split=(97 if w == 33 else 0))
                    out.append((path, want.astype(np.uint8)))
    # a picture large enough for long matches, several deflate blocks and every filter on real content
    yy, xx = np.mgrid[0:301, 0:517]
    big = np.stack([(xx * 3 + yy) & 255, (xx ^ yy) & 255, ((xx * yy) >> 3) & 255], axis=-1)
    big[100:200, 50:400] = (200, 30, 77)
    path = str(tmp_path / "big.png")
    _png_encode(path, big, 2, 8)
    out.append((path, big.astype(np.uint8)))
    return out


def _load_image(path):
    w, h = C.c_uint32(), C.c_uint32()
    if abi.host().mrth_load_image(path.encode(), C.byref(w), C.byref(h), None, 0) != 0:
        return None
    px = np.zeros((h.value, w.value, 3), np.float32)
    assert abi.host().mrth_load_image(path.encode(), C.byref(w), C.byref(h), px.ctypes.data_as(abi.f32p), px.size) == 0
    return px


def test_png_reader_all_formats(built, tmp_path):
    """image_texture on PNG files of every colour type and bit depth, plain and Adam7-interlaced,
    all five scanline filters, stored / fixed / dynamic deflate blocks, IDAT split over many chunks:
    texels = (byte of lodepng's RGBA8 conversion) / 255.f (image.cpp:30-34)."""
    for path, want in _png_cases(tmp_path):
        got = _load_image(path)
        assert got is not None, path
        assert np.array_equal(got, want.astype(np.float32) / np.float32(255.0)), path
    # damaged files are refused, not half-read
    data = open(path, "rb").read()
    for bad in (data[:len(data) // 2], data[:8] + b"\0" * 40, b"P6 2 2"):
        p = str(tmp_path / "bad.png")
        open(p, "wb").write(bad)
        assert _load_image(p) is None
    assert _load_image(str(tmp_path / "missing.png")) is None


@pytest.mark.ref
def test_png_reader_against_the_references_lodepng(built, tmp_path):
    """Container only: image_texture's texels equal what the reference's loadPng (lodepng::decode,
    image.cpp:13-40) returns, on the generated files of every format and on the reference's own
    texture files."""
    lib = orc.ref()
    import glob
    paths = [p for p, _ in _png_cases(tmp_path)]
    paths += sorted(glob.glob("/root/reference/assets/*.png"))[:6] + ["/root/reference/assets/pf-s125-ake7011-a.png", "/root/reference/assets/container2.png"]
    for path in paths:
        got = _load_image(path)
        assert got is not None, path
        ww, hh = C.c_int(), C.c_int()
        px = np.zeros_like(got)
        assert lib.ref_load_png(path.encode(), C.byref(ww), C.byref(hh), orc._p(px), px.size) == 0, path
        assert (hh.value, ww.value) == got.shape[:2]
        assert np.array_equal(got, px), path


def _flat_digest_script():
    return ("import hashlib, sys; sys.path.insert(0, %r)\n"
            "from raytracer_amd import scenes\n"
            "sc = scenes.bunny_ibl(64, 64, env=None)  # 69 451 triangles: node jobs, axis jobs and serial subtrees\n"
            "tree = sc.initKdTree(int(sys.argv[1]))\n"
            "for layout in (0, 1):\n"
            "    n, r = tree.flatten(layout).nodes()\n"
            "    print(hashlib.sha256(n.tobytes() + r.tobytes()).hexdigest())\n") % ROOT


def test_tree_and_flatten_do_not_depend_on_the_thread_count(built):
    """The builder's thread pool and flatten's parallel subtree writes give the same bytes whatever the
    number of threads (1, 3 and 8 builder threads; OMP_NUM_THREADS 1 and 8 for flatten)."""
    import subprocess
    import sys
    out = set()
    for threads, omp in ((1, "1"), (3, "8"), (8, "2")):
        env = dict(os.environ, OMP_NUM_THREADS=omp)
        r = subprocess.run([sys.executable, "-c", _flat_digest_script(), str(threads)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        out.add(tuple(l for l in r.stdout.split() if len(l) == 64))
    assert len(out) == 1 and len(next(iter(out))) == 2, out
