"""Test-side loaders for the CPU oracle (oracle/libmrt_oracle.so) and, in the build container only,
the reference's own translation units (oracle/_ref/libmrt_ref.so).  TEST INFRASTRUCTURE: nothing
under raytracer_amd/ imports this module.
"""
import ctypes as C
import os

import numpy as np

from raytracer_amd import abi
from raytracer_amd.host import tree_dump

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "libmrt_oracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libmrt_ref.so")

_orc = None
_ref = None
vp = C.c_void_p


def _p(a):
    return a.ctypes.data_as(vp) if a is not None else None


def oracle():
    global _orc
    if _orc is None:
        lib = C.CDLL(ORACLE_SO)
        lib.orc_build.restype = vp
        lib.orc_build.argtypes = [C.POINTER(abi.SceneDesc)]
        lib.orc_tree_free.argtypes = [vp]
        lib.orc_tree_counts.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
        lib.orc_tree_flatten.argtypes = [vp] * 11
        lib.orc_tree_to_abi.argtypes = [vp] * 7
        lib.orc_intersect.argtypes = [C.POINTER(abi.SceneDesc), C.c_int64] + [vp] * 4 + [C.c_int] + [vp] * 7 + [C.POINTER(abi.TraceStats), C.c_int]
        lib.orc_object_intersect.argtypes = [C.POINTER(abi.SceneDesc), C.c_int, C.c_int64] + [vp] * 4 + [C.c_int] + [vp] * 7
        lib.orc_intersect_aabb.argtypes = [C.c_int64] + [vp] * 9
        lib.orc_transform.argtypes = [C.c_int, vp, C.POINTER(abi.Transform)]
        lib.orc_sphere_from_transform.argtypes = [C.POINTER(abi.Transform), vp]
        lib.orc_light_shape_sample.argtypes = [C.POINTER(abi.SceneDesc), C.c_int, C.c_int64] + [vp] * 7
        lib.orc_light_shape_area.argtypes = [C.POINTER(abi.SceneDesc), C.c_int]
        lib.orc_light_shape_area.restype = C.c_float
        lib.orc_light_shape_pdf.argtypes = [C.POINTER(abi.SceneDesc), C.c_int, C.c_int64] + [vp] * 4
        lib.orc_beckmann.argtypes = [C.c_int64] + [vp] * 3
        lib.orc_fresnel.argtypes = [C.c_int64] + [vp] * 4
        lib.orc_smith.argtypes = [C.c_int64] + [vp] * 6
        lib.orc_bsdf.argtypes = [C.c_int64] + [vp] * 2
        lib.orc_schlick_f.argtypes = [C.c_int64] + [vp] * 6
        lib.orc_material_eval.argtypes = [C.POINTER(abi.Material), C.c_int64] + [vp] * 5
        lib.orc_rng_u32.argtypes = [C.c_uint64, C.c_int64, vp]
        lib.orc_rng_float.argtypes = [C.c_uint64, C.c_int64, vp]
        lib.orc_rng_bounded.argtypes = [C.c_uint64, C.c_int64, vp, vp]
        lib.orc_stratified_stream.argtypes = [C.c_int] * 6 + [vp]
        lib.orc_uniform_sample_triangle.argtypes = [C.c_int64, vp, vp]
        lib.orc_concentric_disk.argtypes = [C.c_int64, vp, vp]
        lib.orc_power_heuristic.argtypes = [C.c_int64, vp, vp, vp]
        lib.orc_dist1d_discrete.argtypes = [vp, C.c_int, C.c_int64, vp, vp, vp]
        lib.orc_dist2d_pdf.argtypes = [vp, C.c_int, C.c_int, C.c_int64, vp, vp]
        lib.orc_camera.argtypes = [vp, vp, vp] + [C.c_float] * 5 + [C.POINTER(abi.Camera)]
        lib.orc_camera_rays.argtypes = [C.POINTER(abi.Camera), C.c_int64] + [vp] * 4
        lib.orc_sample_streams.argtypes = [C.c_uint32] * 5 + [vp] * 3
        lib.orc_texture_value.argtypes = [C.POINTER(abi.SceneDesc), C.c_int, C.c_int64, vp, vp]
        lib.orc_random_dir_replay.argtypes = [C.c_int64, vp, vp, vp]
        lib.orc_coordinate_system.argtypes = [C.c_int64, vp, vp, vp]
        lib.orc_uniform_light_distribution.argtypes = [C.c_int, C.c_int64, vp, vp, vp]
        lib.orc_face_normals.argtypes = [C.c_int64, vp, C.c_int, vp, vp, vp]
        lib.orc_render.argtypes = [C.POINTER(abi.SceneDesc), C.POINTER(abi.RenderParams), C.c_int, vp, vp,
                                   C.POINTER(abi.RenderStats), C.POINTER(abi.TraceStats), vp, C.c_int]
        _orc = lib
    return _orc


def have_ref():
    return os.path.exists(REF_SO)


def ref():
    global _ref
    if _ref is None:
        lib = C.CDLL(REF_SO)
        lib.ref_xf_new.restype = vp
        lib.ref_xf_free.argtypes = [vp]
        lib.ref_xf_scale.argtypes = [vp] + [C.c_float] * 3
        lib.ref_xf_rotate.argtypes = [vp] + [C.c_float] * 4
        lib.ref_xf_translate.argtypes = [vp] + [C.c_float] * 3
        lib.ref_xf_get.argtypes = [vp] * 4
        lib.ref_scene_new.restype = vp
        lib.ref_add_triangle.argtypes = [vp] * 7
        lib.ref_add_sphere.argtypes = [vp] * 4
        lib.ref_add_plane.argtypes = [vp] * 3
        lib.ref_build.restype = vp
        lib.ref_build.argtypes = [vp]
        lib.ref_tree_counts.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
        lib.ref_tree_flatten.argtypes = [vp] * 11
        lib.ref_intersect.argtypes = [vp, vp, C.c_int64] + [vp] * 13
        lib.ref_object_intersect.argtypes = [vp, C.c_int, C.c_int64] + [vp] * 12
        lib.ref_intersect_aabb.argtypes = [C.c_int64] + [vp] * 9
        lib.ref_object_sample.argtypes = [vp, C.c_int, C.c_int64] + [vp] * 7
        lib.ref_object_area.argtypes = [vp, C.c_int]
        lib.ref_object_area.restype = C.c_float
        lib.ref_object_pdf.argtypes = [vp, C.c_int, C.c_int64] + [vp] * 4
        lib.ref_beckmann.argtypes = [C.c_int64] + [vp] * 3
        lib.ref_fresnel.argtypes = [C.c_int64] + [vp] * 4
        lib.ref_smith.argtypes = [C.c_int64] + [vp] * 6
        lib.ref_bsdf.argtypes = [C.c_int64] + [vp] * 2
        lib.ref_schlick_f.argtypes = [C.c_int64] + [vp] * 6
        lib.ref_rng_u32.argtypes = [C.c_uint64, C.c_int64, vp]
        lib.ref_rng_float.argtypes = [C.c_uint64, C.c_int64, vp]
        lib.ref_rng_bounded.argtypes = [C.c_uint64, C.c_int64, vp, vp]
        lib.ref_stratified_stream.argtypes = [C.c_int] * 6 + [vp]
        lib.ref_uniform_sample_triangle.argtypes = [C.c_int64, vp, vp]
        lib.ref_concentric_disk.argtypes = [C.c_int64, vp, vp]
        lib.ref_power_heuristic.argtypes = [C.c_int64, vp, vp, vp]
        lib.ref_dist1d_discrete.argtypes = [vp, C.c_int, C.c_int64, vp, vp, vp]
        lib.ref_dist2d_pdf.argtypes = [vp, C.c_int, C.c_int, C.c_int64, vp, vp]
        lib.ref_camera_new.restype = vp
        lib.ref_camera_new.argtypes = [vp, vp, vp] + [C.c_float] * 5
        lib.ref_camera_rays.argtypes = [vp, C.c_int64] + [vp] * 4
        lib.ref_camera_params.argtypes = [vp, vp]
        lib.ref_checker_value.argtypes = [vp, vp, vp, C.c_int64, vp, vp]
        lib.ref_image_value.argtypes = [C.c_char_p, C.c_int, C.c_int, vp, vp, C.c_int64, vp, vp]
        lib.ref_random_dir.argtypes = [C.c_int64, vp, vp, vp, vp]
        lib.ref_coordinate_system.argtypes = [C.c_int64, vp, vp, vp]
        lib.ref_uniform_light_distribution.argtypes = [C.c_int, C.c_int64, vp, vp, vp]
        lib.ref_obj_load.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32),
                                     C.POINTER(C.c_int32), vp, vp, vp, vp, vp]
        lib.ref_adapter_flatten.restype = vp
        lib.ref_adapter_flatten.argtypes = [vp, vp]
        lib.ref_adapter_desc.restype = C.POINTER(abi.SceneDesc)
        lib.ref_adapter_desc.argtypes = [vp]
        lib.ref_adapter_free.argtypes = [vp]
        lib.ref_scene_set_camera.argtypes = [vp, vp]
        lib.ref_material_set_checker.argtypes = [vp, C.c_int, vp, vp, vp]
        lib.ref_material_set_image.argtypes = [vp, C.c_int, C.c_char_p, vp]
        lib.ref_load_png.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), vp, C.c_int64]
        lib.ref_save_image.argtypes = [C.c_char_p, C.c_int, C.c_int, vp]
        lib.ref_face_normals.argtypes = [C.c_int64, vp, C.c_int, vp, vp, vp]
        _ref = lib
    return _ref


# ---------------------------------------------------------------------------------------- transforms
def apply_ops(handle, ops, scale, rotate, translate):
    for op in ops:
        if op[0] == "scale":
            scale(handle, *[float(x) for x in op[1:4]])
        elif op[0] == "rotate":
            rotate(handle, *[float(x) for x in op[1:5]])
        else:
            translate(handle, *[float(x) for x in op[1:4]])


def ops_array(ops):
    rows = []
    for op in ops:
        kind = {"scale": 0, "rotate": 1, "translate": 2}[op[0]]
        vals = list(op[1:]) + [0.0] * (4 - len(op[1:]))
        rows.append([kind] + [float(v) for v in vals])
    return np.asarray(rows, np.float32).reshape(-1, 5)


def oracle_transform(ops):
    a = ops_array(ops)
    t = abi.Transform()
    oracle().orc_transform(len(ops), _p(a), C.byref(t))
    return np.array(t.m, np.float32), np.array(t.inv, np.float32), np.array(t.t, np.float32)


def ref_transform_handle(ops):
    lib = ref()
    h = lib.ref_xf_new()
    apply_ops(h, ops, lib.ref_xf_scale, lib.ref_xf_rotate, lib.ref_xf_translate)
    return h


def ref_transform(ops):
    h = ref_transform_handle(ops)
    m, inv, t = np.zeros(9, np.float32), np.zeros(9, np.float32), np.zeros(3, np.float32)
    ref().ref_xf_get(h, _p(m), _p(inv), _p(t))
    return m, inv, t


# ---------------------------------------------------------------------------------------- trees
def oracle_tree(desc):
    lib = oracle()
    t = lib.orc_build(desc)
    nn, nr, dp = C.c_int64(), C.c_int64(), C.c_int32()
    lib.orc_tree_counts(t, C.byref(nn), C.byref(nr), C.byref(dp))
    dump = tree_dump(lambda *ptrs: lib.orc_tree_flatten(t, *ptrs), nn.value, nr.value)
    dump["depth"] = dp.value
    lib.orc_tree_free(t)
    return dump


def canonical_from_abi(nodes, refs):
    """Pre-order (axis, split bits, leaf refs) walk of an ABI node array -> comparable tuple of arrays."""
    axis, split, leaf_refs, counts = [], [], [], []
    stack = [0] if len(nodes) else []
    while stack:
        n = stack.pop()
        w0, w1 = int(nodes[n, 0]), int(nodes[n, 1])
        if (w1 & 3) == 3:
            axis.append(-1)
            split.append(0)
            cnt = w1 >> 2
            counts.append(cnt)
            leaf_refs.extend(refs[w0:w0 + cnt].tolist())
        else:
            axis.append(w1 & 3)
            split.append(w0)
            counts.append(0)
            left = w1 >> 2
            stack.append(left + 1)
            stack.append(left)
    return (np.asarray(axis, np.int32), np.asarray(split, np.uint32), np.asarray(counts, np.int32),
            np.asarray(leaf_refs, np.int32))


def canonical_from_dump(d):
    split = np.where(d["leaf"] == 1, np.uint32(0), d["split"].view(np.uint32))
    return d["axis"].astype(np.int32), split.astype(np.uint32), d["ref_cnt"].astype(np.int32), d["refs"].astype(np.int32)


def tree_digest(canon):
    import hashlib
    h = hashlib.sha256()
    for a in canon:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def xf_ops_for(flat, scene):
    """transform index of the flat description -> the operation list that built it."""
    d = flat.desc.contents
    out = {}
    for xf in scene.transforms:
        m, inv, t = xf.get()
        for i in range(d.n_transforms):
            tt = d.transforms[i]
            if (np.array_equal(m.view(np.uint32), np.array(tt.m, np.float32).view(np.uint32)) and
                    np.array_equal(t.view(np.uint32), np.array(tt.t, np.float32).view(np.uint32))):
                out[i] = list(xf.ops)
    return out


# ---------------------------------------------------------------------------------------- reference scene
class RefScene:
    """Mirror of a flat scene description inside the reference library.

    xf_ops maps a transform index of the description to the list of operations that built it
    (index 0, the identity, needs none); the reference Transform is rebuilt from the operations so
    that its matrices come from the reference's own arithmetic.
    """

    def __init__(self, desc, xf_ops=None):
        lib = ref()
        d = desc.contents
        self.scene = lib.ref_scene_new()
        xf_ops = xf_ops or {}
        handles = {}
        n = d.n_objects
        kind = np.ctypeslib.as_array(d.obj_kind, shape=(n,))
        mat = np.ctypeslib.as_array(d.obj_material, shape=(n,))
        xfi = np.ctypeslib.as_array(d.obj_transform, shape=(n,))
        tri_p = np.ctypeslib.as_array(d.tri_p, shape=(n, 9))
        tri_n = np.ctypeslib.as_array(d.tri_n, shape=(n, 9))
        tri_uv = np.ctypeslib.as_array(d.tri_uv, shape=(n, 6))
        zero3 = np.zeros(3, np.float32)
        for i in range(n):
            x = int(xfi[i])
            if x not in handles:
                handles[x] = ref_transform_handle(xf_ops.get(x, []))
                m, inv, t = np.zeros(9, np.float32), np.zeros(9, np.float32), np.zeros(3, np.float32)
                lib.ref_xf_get(handles[x], _p(m), _p(inv), _p(t))
                tt = d.transforms[x]
                assert np.array_equal(m.view(np.uint32), np.array(tt.m, np.float32).view(np.uint32)), "transform %d differs" % x
                assert np.array_equal(inv.view(np.uint32), np.array(tt.inv, np.float32).view(np.uint32)), "inverse %d differs" % x
            em = np.array(d.materials[int(mat[i])].emission, np.float32)
            if kind[i] == abi.MRT_TRIANGLE:
                lib.ref_add_triangle(self.scene, _p(np.ascontiguousarray(tri_p[i])), _p(np.ascontiguousarray(tri_n[i])),
                                     _p(zero3), _p(np.ascontiguousarray(tri_uv[i])), handles[x], _p(em))
            elif kind[i] == abi.MRT_SPHERE:
                lib.ref_add_sphere(self.scene, handles[x], _p(em), None)
            else:
                lib.ref_add_plane(self.scene, handles[x], _p(em))
        self.tree = lib.ref_build(self.scene)

    def dump(self):
        lib = ref()
        nn, nr, dp = C.c_int64(), C.c_int64(), C.c_int32()
        lib.ref_tree_counts(self.tree, C.byref(nn), C.byref(nr), C.byref(dp))
        d = tree_dump(lambda *ptrs: lib.ref_tree_flatten(self.tree, *ptrs), nn.value, nr.value)
        d["depth"] = dp.value
        return d

    def intersect(self, o, d, tmin, tmax, shadow):
        n = o.shape[0]
        out = trace_out(n)
        sh = np.full(n, 1 if shadow else 0, np.uint8)
        ref().ref_intersect(self.scene, self.tree, n, _p(o), _p(d), _p(tmin), _p(tmax), _p(sh), _p(out["hit"]),
                            _p(out["prim"]), _p(out["t"]), _p(out["u"]), _p(out["v"]), _p(out["pos"]), _p(out["nrm"]),
                            _p(out["outside"]))
        return out


def trace_out(n):
    return {"hit": np.zeros(n, np.uint8), "prim": np.zeros(n, np.int32), "t": np.zeros(n, np.float32),
            "u": np.zeros(n, np.float32), "v": np.zeros(n, np.float32), "pos": np.zeros((n, 3), np.float32),
            "nrm": np.zeros((n, 3), np.float32), "outside": np.zeros(n, np.uint8)}


def oracle_intersect(desc, o, d, tmin, tmax, shadow, stats=False, threads=8):
    n = o.shape[0]
    out = trace_out(n)
    ts = abi.TraceStats()
    oracle().orc_intersect(desc, n, _p(o), _p(d), _p(tmin), _p(tmax), 1 if shadow else 0, _p(out["prim"]), _p(out["t"]),
                           _p(out["u"]), _p(out["v"]), _p(out["pos"]), _p(out["nrm"]), _p(out["outside"]), C.byref(ts), threads)
    out["hit"] = (out["prim"] >= 0).astype(np.uint8)
    if stats:
        return out, ts
    return out


def oracle_render(desc, params, sampler_mode=0, threads=8, per_sample=False):
    h, w = params.height, params.width
    img = np.zeros((h, w, 3), np.float32)
    rs, ts = abi.RenderStats(), (abi.TraceStats * 3)()
    rays = np.zeros(3, np.uint64)
    ps = None
    if per_sample:
        rows = (params.row_end or h) - params.row_begin
        ns = (params.sample_end or params.spp_x * params.spp_y) - params.sample_begin
        ps = np.zeros((rows, w, ns, 3), np.float32)
    oracle().orc_render(desc, C.byref(params), sampler_mode, _p(img), _p(ps), C.byref(rs), ts, _p(rays), threads)
    rs.class_rays = rays
    return img, rs, ts, ps


# ---------------------------------------------------------------------------------------- ray generators
def camera_rays(desc, width, height, seed=0, jitter=True):
    """One ray per pixel through the scene's SimpleCamera (oracle restatement of get_ray)."""
    rng = np.random.default_rng(seed)
    n = width * height
    ys, xs = np.divmod(np.arange(n), width)
    st = np.stack([xs, ys], axis=1).astype(np.float32)
    if jitter:
        st += rng.random((n, 2), dtype=np.float32)
    else:
        st += 0.5
    lens = rng.random((n, 2), dtype=np.float32)
    o, d = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
    cam = desc.contents.camera
    oracle().orc_camera_rays(C.byref(cam), n, _p(np.ascontiguousarray(st)), _p(lens), _p(o), _p(d))
    return o, d


def random_rays(desc, n, seed=1):
    """Incoherent rays: origins inside the padded root box, uniform directions; also a share of
    axis-parallel directions and origins snapped onto split planes."""
    rng = np.random.default_rng(seed)
    dd = desc.contents
    lo, hi = np.array(dd.root_min, np.float32), np.array(dd.root_max, np.float32)
    o = (lo + (hi - lo) * rng.random((n, 3), dtype=np.float32)).astype(np.float32)
    v = rng.normal(size=(n, 3)).astype(np.float32)
    v /= np.linalg.norm(v, axis=1, keepdims=True).astype(np.float32)
    k = n // 16
    for a in range(3):  # axis-parallel
        v[a * k:(a + 1) * k] = 0
        v[a * k:(a + 1) * k, a] = np.where(rng.random(k) < 0.5, -1.0, 1.0)
    # origins exactly on split planes of the top of the tree
    nn = dd.n_nodes
    if nn:
        nodes = np.ctypeslib.as_array(C.cast(dd.nodes, abi.u32p), shape=(nn * 2,)).reshape(nn, 2)
        inner = np.nonzero((nodes[:min(nn, 4096), 1] & 3) != 3)[0]
        if len(inner):
            pick = rng.choice(inner, size=k)
            ax = (nodes[pick, 1] & 3).astype(np.int64)
            sp = nodes[pick, 0].copy().view(np.float32)
            o[3 * k + np.arange(k), ax] = sp
    return np.ascontiguousarray(o), np.ascontiguousarray(v.astype(np.float32))
