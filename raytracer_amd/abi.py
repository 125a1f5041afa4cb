"""ctypes view of include/mrt.h and include/mrt_host_c.h.

The product is the two shared libraries; this module only loads them and mirrors their structs so
that the Python tests and the bench driver can call the C-ABI.  There is no Python fallback: if a
library is missing the import fails.
"""
import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
HIP_SO = os.path.join(_PKG, "libmrt_hip.so")
HOST_SO = os.path.join(_PKG, "libmrt_host.so")

MRT_OK = 0
MRT_SPHERE, MRT_PLANE, MRT_TRIANGLE = 1, 2, 3
MRT_MAT_DIFFUSE, MRT_MAT_SPECULAR, MRT_MAT_TRANSPARENT = 0, 1, 2
MRT_LIGHT_SHAPE, MRT_LIGHT_IBL = 0, 1
MRT_TEX_IMAGE, MRT_TEX_CHECKER = 0, 1
MRT_BUF_HOST, MRT_BUF_DEVICE = 0, 1
MRT_CLOSEST_HIT, MRT_ANY_HIT = 0, 1
MRT_RENDER_TRACE_STATS = 1
MRT_RENDER_SERIAL = 2
MRT_RENDER_NO_KIND_BUCKETS = 4

f32p = C.POINTER(C.c_float)
u32p = C.POINTER(C.c_uint32)
i32p = C.POINTER(C.c_int32)
u8p = C.POINTER(C.c_uint8)


class Transform(C.Structure):
    _fields_ = [("m", C.c_float * 9), ("inv", C.c_float * 9), ("t", C.c_float * 3)]


class Material(C.Structure):
    _fields_ = [("albedo", C.c_float * 3), ("emission", C.c_float * 3), ("ior", C.c_float),
                ("roughness", C.c_float), ("metalness", C.c_float), ("type", C.c_int32), ("texture", C.c_int32)]


class Texture(C.Structure):
    _fields_ = [("kind", C.c_int32), ("width", C.c_uint32), ("height", C.c_uint32), ("texel_offset", C.c_uint64),
                ("even", C.c_float * 3), ("odd", C.c_float * 3), ("xf", Transform)]


class KdNode(C.Structure):
    _fields_ = [("w0", C.c_uint32), ("w1", C.c_uint32)]


class Light(C.Structure):
    _fields_ = [("kind", C.c_int32), ("color", C.c_float * 3), ("shape_kind", C.c_int32), ("transform", C.c_uint32),
                ("tri_p", C.c_float * 9), ("tri_n", C.c_float * 9), ("sphere", C.c_float * 4)]


class Camera(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("near_top_left", C.c_float * 3), ("dx_pixel", C.c_float * 3),
                ("dy_pixel", C.c_float * 3), ("x_dir", C.c_float * 3), ("up", C.c_float * 3), ("lens_radius", C.c_float)]


class SceneDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("n_objects", C.c_uint32),
        ("obj_kind", u8p), ("obj_material", u32p), ("obj_transform", u32p),
        ("tri_p", f32p), ("tri_n", f32p), ("tri_uv", f32p), ("sphere", f32p),
        ("n_transforms", C.c_uint32), ("transforms", C.POINTER(Transform)),
        ("n_materials", C.c_uint32), ("materials", C.POINTER(Material)),
        ("n_textures", C.c_uint32), ("textures", C.POINTER(Texture)),
        ("n_texels", C.c_uint64), ("texels", f32p),
        ("n_nodes", C.c_uint32), ("nodes", C.POINTER(KdNode)),
        ("n_refs", C.c_uint32), ("refs", u32p),
        ("root_min", C.c_float * 3), ("root_max", C.c_float * 3),
        ("n_out_of_tree", C.c_uint32), ("out_of_tree", u32p),
        ("n_lights", C.c_uint32), ("lights", C.POINTER(Light)),
        ("env_light", C.c_int32), ("env_width", C.c_uint32), ("env_height", C.c_uint32),
        ("env_rgb", f32p), ("env_inv", C.c_float * 9),
        ("camera", Camera), ("has_camera", C.c_uint32),
    ]


class Rays(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ox", "oy", "oz", "dx", "dy", "dz", "tmin", "tmax")]


class Hits(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("prim", "t", "u", "v")]


class TraceStats(C.Structure):
    _fields_ = [("inner", C.c_uint64), ("leaves", C.c_uint64), ("refs", C.c_uint64), ("prim_tests", C.c_uint64)]


class RenderParams(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("width", "height", "spp_x", "spp_y", "sample_begin", "sample_end",
                                          "max_depth", "seed", "row_begin", "row_end", "paths_per_pass", "flags",
                                          "tile_step", "tile_phase")]


class RenderStats(C.Structure):
    _fields_ = [("rays_closest", C.c_uint64), ("rays_any", C.c_uint64), ("paths", C.c_uint64),
                ("extend_ms", C.c_double), ("shade_ms", C.c_double), ("connect_ms", C.c_double), ("generate_ms", C.c_double),
                ("trace", TraceStats * 3), ("rays_mis", C.c_uint64), ("connect_mis_ms", C.c_double),
                ("passes", C.c_uint64)]


class DeviceInfo(C.Structure):
    _fields_ = [("device", C.c_int32), ("compute_units", C.c_int32), ("xcd_mask", C.c_uint32), ("kd_tree_depth", C.c_uint32)]


_hip = None
_host = None


def hip():
    """libmrt_hip.so (the C-ABI).  Raises if the extension has not been built."""
    global _hip
    if _hip is None:
        if not os.path.exists(HIP_SO):
            raise ImportError(HIP_SO + " is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        lib = C.CDLL(HIP_SO, mode=C.RTLD_GLOBAL)
        lib.mrt_last_error.restype = C.c_char_p
        lib.mrt_version.restype = C.c_int32
        lib.mrt_scene_create.argtypes = [C.POINTER(SceneDesc), C.c_int, C.POINTER(C.c_void_p)]
        lib.mrt_scene_create.restype = C.c_int32
        lib.mrt_scene_destroy.argtypes = [C.c_void_p]
        lib.mrt_scene_destroy.restype = None
        lib.mrt_intersect.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(Rays), C.POINTER(Hits), C.c_uint32, C.c_uint32,
                                      C.c_void_p, C.POINTER(TraceStats)]
        lib.mrt_intersect.restype = C.c_int32
        lib.mrt_intersect_noalloc.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(Rays), C.POINTER(Hits), C.c_uint32, C.c_void_p]
        lib.mrt_intersect_noalloc.restype = C.c_int32
        lib.mrt_render.argtypes = [C.c_void_p, C.POINTER(RenderParams), C.c_void_p, C.c_uint32, C.c_void_p,
                                   C.POINTER(RenderStats)]
        lib.mrt_render.restype = C.c_int32
        lib.mrt_resolve.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
        lib.mrt_resolve.restype = C.c_int32
        lib.mrt_scene_device_info.argtypes = [C.c_void_p, C.POINTER(DeviceInfo)]
        lib.mrt_scene_device_info.restype = C.c_int32
        lib.mrt_reduce.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int32, C.c_uint64, C.c_int32]
        lib.mrt_reduce.restype = C.c_int32
        if hasattr(lib, "mrt_reduce_async"):
            lib.mrt_reduce_async.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int32, C.c_uint64, C.c_int32, C.c_void_p]
            lib.mrt_reduce_async.restype = C.c_int32
            lib.mrt_scene_trim.argtypes = [C.c_void_p]
            lib.mrt_scene_trim.restype = C.c_int32
        if hasattr(lib, "mrt_debug_camera_choice"):
            lib.mrt_debug_camera_choice.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_float)]
            lib.mrt_debug_camera_choice.restype = C.c_int32
        lib.mrt_render_multi.argtypes = [C.POINTER(C.c_void_p), C.c_int32, C.POINTER(RenderParams), C.c_void_p, C.POINTER(RenderStats)]
        lib.mrt_render_multi.restype = C.c_int32
        lib.mrt_debug_eval.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, f32p, C.c_uint32, f32p, C.c_uint32]
        lib.mrt_debug_eval.restype = C.c_int32
        if hasattr(lib, "mrt_debug_relayed"):  # (absent from builds of earlier revisions that tools/ab*.sh compare with)
            lib.mrt_debug_relayed.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
            lib.mrt_debug_relayed.restype = C.c_int32
        _hip = lib
    return _hip


def host():
    """libmrt_host.so (scene set-up, kd-tree build, flatten, HIPRenderer)."""
    global _host
    if _host is None:
        hip()
        if not os.path.exists(HOST_SO):
            raise ImportError(HOST_SO + " is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        lib = C.CDLL(HOST_SO)
        vp, ci, cf = C.c_void_p, C.c_int, C.c_float
        sig = {
            "mrth_last_error": (C.c_char_p, []),
            "mrth_transform_new": (vp, []),
            "mrth_transform_free": (None, [vp]),
            "mrth_transform_scale": (None, [vp, cf, cf, cf]),
            "mrth_transform_rotate": (None, [vp, cf, cf, cf, cf]),
            "mrth_transform_translate": (None, [vp, cf, cf, cf]),
            "mrth_transform_get": (None, [vp, C.POINTER(Transform)]),
            "mrth_scene_new": (vp, []),
            "mrth_scene_free": (None, [vp]),
            "mrth_create_material": (ci, [vp, ci]),
            "mrth_material_set": (ci, [vp, ci, C.POINTER(Material)]),
            "mrth_material_get": (ci, [vp, ci, C.POINTER(Material)]),
            "mrth_material_set_checker": (ci, [vp, ci, f32p, f32p, vp]),
            "mrth_material_set_image": (ci, [vp, ci, C.c_uint32, C.c_uint32, f32p, vp]),
            "mrth_materials_share_texture": (ci, [vp, ci, ci, ci]),
            "mrth_shuffle_objects": (None, [vp, C.c_uint64]),
            "mrth_n_objects": (ci, [vp]),
            "mrth_n_materials": (ci, [vp]),
            "mrth_n_lights": (ci, [vp]),
            "mrth_add_obj_file": (ci, [vp, C.c_char_p, ci, vp, cf, ci]),
            "mrth_add_triangle": (ci, [vp, f32p, f32p, f32p, f32p, vp, ci]),
            "mrth_add_sphere": (ci, [vp, vp, ci]),
            "mrth_add_plane": (ci, [vp, vp, ci]),
            "mrth_add_sphere_light": (ci, [vp, vp, ci, f32p]),
            "mrth_add_triangle_light": (ci, [vp, f32p, f32p, f32p, vp, ci, f32p]),
            "mrth_add_ibl_file": (ci, [vp, C.c_char_p, ci]),
            "mrth_add_ibl_pixels": (ci, [vp, ci, ci, f32p, ci]),
            "mrth_set_simple_camera": (None, [vp, f32p, f32p, f32p, cf, cf, cf, cf, cf]),
            "mrth_build_kdtree": (vp, [vp, ci]),
            "mrth_tree_free": (None, [vp]),
            "mrth_tree_counts": (None, [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
            "mrth_tree_dump": (None, [vp] + [vp] * 10),
            "mrth_flatten": (vp, [vp, vp, ci]),
            "mrth_flat_desc": (C.POINTER(SceneDesc), [vp]),
            "mrth_flat_depth": (C.c_uint32, [vp]),
            "mrth_flat_free": (None, [vp]),
            "mrth_intersect_kdtree": (ci, [vp, vp, f32p, f32p, C.c_float, C.c_float, ci, f32p]),
            "mrth_hip_render": (ci, [vp, C.c_uint32, C.c_uint32, ci, ci, ci, C.c_uint32, ci, f32p,
                                     C.POINTER(RenderStats), C.POINTER(C.c_double)]),
            "mrth_save_png": (ci, [C.c_char_p, f32p, C.c_uint32, C.c_uint32]),
            "mrth_parse_scene": (vp, [ci, ci, ci, C.c_char_p]),
            "mrth_load_scene_file": (vp, [C.c_char_p, ci, ci, C.c_char_p]),
            "mrth_load_image": (ci, [C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), f32p, C.c_uint64]),
            "mrth_material_set_image_file": (ci, [vp, ci, C.c_char_p, vp]),
            "mrth_hip_render_devices": (ci, [vp, C.c_uint32, C.c_uint32, ci, ci, ci, C.c_uint32, C.POINTER(C.c_int), ci, f32p,
                                             C.POINTER(RenderStats), C.POINTER(C.c_double)]),
            "mrth_hip_render_progressive": (ci, [vp, C.c_uint32, C.c_uint32, ci, ci, ci, C.c_uint32, ci, C.c_uint32, ci,
                                                 C.c_char_p, f32p, C.POINTER(RenderStats)]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _host = lib
    return _host
