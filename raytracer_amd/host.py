"""Python handles over the host library and the C-ABI (thin: every method is one native call).

Names follow the reference's API: Transform.scale/rotate/translate (defines.h:67-148),
Scene.CreateMaterial (scene.h:33), addObjectsFromFile (example_scenes.cpp:22), setSimpleCamera
(scene.cpp:151), initKdTree (kdtree.h:35), intersectKdTree (kdtree.h:34, batched).
"""
import ctypes as C

import numpy as np

from . import abi


def _f32(a, n=None):
    arr = np.ascontiguousarray(np.asarray(a, dtype=np.float32).reshape(-1))
    if n is not None and arr.size != n:
        raise ValueError("expected %d floats, got %d" % (n, arr.size))
    return arr


def _fp(arr):
    return arr.ctypes.data_as(abi.f32p)


class MrtError(RuntimeError):
    pass


class Transform:
    def __init__(self):
        self._h = abi.host().mrth_transform_new()
        self.ops = []  # the operations applied so far, in order (lets a test replay them elsewhere)

    def __del__(self):
        if getattr(self, "_h", None) and abi is not None and getattr(abi, "host", None):  # not during interpreter shutdown
            abi.host().mrth_transform_free(self._h)
            self._h = None

    def scale(self, sx, sy, sz):
        abi.host().mrth_transform_scale(self._h, sx, sy, sz)
        self.ops.append(("scale", sx, sy, sz))
        return self

    def rotate(self, axis, degrees):
        abi.host().mrth_transform_rotate(self._h, axis[0], axis[1], axis[2], degrees)
        self.ops.append(("rotate", axis[0], axis[1], axis[2], degrees))
        return self

    def translate(self, p):
        abi.host().mrth_transform_translate(self._h, p[0], p[1], p[2])
        self.ops.append(("translate", p[0], p[1], p[2]))
        return self

    def get(self):
        t = abi.Transform()
        abi.host().mrth_transform_get(self._h, C.byref(t))
        return (np.array(t.m, dtype=np.float32), np.array(t.inv, dtype=np.float32), np.array(t.t, dtype=np.float32))


class Scene:
    """initScene() ... freeScene()."""

    def __init__(self, handle=None):
        self._h = handle if handle is not None else abi.host().mrth_scene_new()
        self.transforms = []  # every Transform handed to this scene (tests replay their ops)

    @classmethod
    def parse(cls, scene_id, width=1200, height=800, asset_dir=None):
        """parseScene(sceneId) over the scene data files of asset_dir/scenes/."""
        lib = abi.host()
        h = lib.mrth_parse_scene(scene_id, width, height, asset_dir.encode() if asset_dir else None)
        if not h:
            raise MrtError(lib.mrth_last_error().decode())
        return cls(h)

    @classmethod
    def load(cls, filename, width, height, asset_dir=None):
        lib = abi.host()
        h = lib.mrth_load_scene_file(filename.encode(), width, height, asset_dir.encode() if asset_dir else None)
        if not h:
            raise MrtError(lib.mrth_last_error().decode())
        return cls(h)

    def _xf(self, xf):
        self.transforms.append(xf)
        return xf._h

    def __del__(self):
        if getattr(self, "_h", None) and abi is not None and getattr(abi, "host", None):  # not during interpreter shutdown
            abi.host().mrth_scene_free(self._h)
            self._h = None

    # Scene::CreateMaterial(COOK_TORRANCE, type) followed by field assignments
    def CreateMaterial(self, mat_type=abi.MRT_MAT_DIFFUSE, albedo=(1, 1, 1), emission=(0, 0, 0), ior=1.0,
                       roughness=0.1, metalness=0.001):
        lib = abi.host()
        mid = lib.mrth_create_material(self._h, mat_type)
        m = abi.Material()
        m.albedo[:] = [float(x) for x in albedo]
        m.emission[:] = [float(x) for x in emission]
        m.ior, m.roughness, m.metalness, m.type, m.texture = ior, roughness, metalness, mat_type, -1
        if lib.mrth_material_set(self._h, mid, C.byref(m)) != 0:
            raise MrtError(lib.mrth_last_error().decode())
        return mid

    def set_checker(self, mat, even, odd, tex_xf=None):
        e, o = _f32(even, 3), _f32(odd, 3)
        abi.host().mrth_material_set_checker(self._h, mat, _fp(e), _fp(o), tex_xf._h if tex_xf else None)

    def set_image_texture(self, mat, rgb, tex_xf=None):
        rgb = np.ascontiguousarray(rgb, dtype=np.float32)
        h, w, _ = rgb.shape
        abi.host().mrth_material_set_image(self._h, mat, w, h, _fp(rgb.reshape(-1)), tex_xf._h if tex_xf else None)

    def set_image_texture_file(self, mat, filename, tex_xf=None):
        """m_texture = new image_texture(filename): a PNG or binary PPM file."""
        if abi.host().mrth_material_set_image_file(self._h, mat, filename.encode(), tex_xf._h if tex_xf else None) != 0:
            raise MrtError(abi.host().mrth_last_error().decode())

    def share_texture(self, from_mat, first, last):
        """Materials first..last use the texture of from_mat (the texels are stored once)."""
        if abi.host().mrth_materials_share_texture(self._h, from_mat, first, last) != 0:
            raise MrtError(abi.host().mrth_last_error().decode())

    def material(self, mid):
        m = abi.Material()
        if abi.host().mrth_material_get(self._h, mid, C.byref(m)) != 0:
            raise MrtError(abi.host().mrth_last_error().decode())
        return m

    @property
    def n_objects(self):
        return abi.host().mrth_n_objects(self._h)

    @property
    def n_materials(self):
        return abi.host().mrth_n_materials(self._h)

    @property
    def n_lights(self):
        return abi.host().mrth_n_lights(self._h)

    def addObjectsFromFile(self, path, default_mat, xf, prescale=1.0, subdivisions=0):
        n = abi.host().mrth_add_obj_file(self._h, str(path).encode(), default_mat, self._xf(xf), prescale, subdivisions)
        if n < 0:
            raise MrtError(abi.host().mrth_last_error().decode())
        return n

    def add_triangle(self, p, n, fn, uv, xf, mat):
        p, n, fn, uv = _f32(p, 9), _f32(n, 9), _f32(fn, 3), _f32(uv, 6)
        return abi.host().mrth_add_triangle(self._h, _fp(p), _fp(n), _fp(fn), _fp(uv), self._xf(xf), mat)

    def add_sphere(self, xf, mat):
        return abi.host().mrth_add_sphere(self._h, self._xf(xf), mat)

    def add_plane(self, xf, mat):
        return abi.host().mrth_add_plane(self._h, self._xf(xf), mat)

    def add_sphere_light(self, xf, mat, color):
        c = _f32(color, 3)
        return abi.host().mrth_add_sphere_light(self._h, self._xf(xf), mat, _fp(c))

    def add_triangle_light(self, p, n, fn, xf, mat, color):
        p, n, fn, c = _f32(p, 9), _f32(n, 9), _f32(fn, 3), _f32(color, 3)
        return abi.host().mrth_add_triangle_light(self._h, _fp(p), _fp(n), _fp(fn), self._xf(xf), mat, _fp(c))

    def add_ibl_file(self, path, also_env=True):
        r = abi.host().mrth_add_ibl_file(self._h, str(path).encode(), 1 if also_env else 0)
        if r < 0:
            raise MrtError(abi.host().mrth_last_error().decode())
        return r

    def add_ibl_pixels(self, rgb, also_env=True):
        rgb = np.ascontiguousarray(rgb, dtype=np.float32)
        h, w, _ = rgb.shape
        return abi.host().mrth_add_ibl_pixels(self._h, w, h, _fp(rgb.reshape(-1)), 1 if also_env else 0)

    def setSimpleCamera(self, position, at, up, fov, w, h, radius, focus):
        a, b, c = _f32(position, 3), _f32(at, 3), _f32(up, 3)
        abi.host().mrth_set_simple_camera(self._h, _fp(a), _fp(b), _fp(c), fov, w, h, radius, focus)

    def initKdTree(self, n_threads=0):
        return KdTree(self, n_threads)

    def hip_render(self, width, height, spp_x, spp_y, max_depth=5, seed=1, device=0, devices=None):
        """HIPRenderer::init + run + destroy (the GPU sibling of CPURenderer).  Returns the linear
        radiance sums [h][w][3], the render statistics and the seconds run() took.  `devices`: a list
        of device indices among which the samples are divided (HIPRenderer::devices)."""
        img = np.zeros((height, width, 3), np.float32)
        st, sec = abi.RenderStats(), C.c_double()
        devs = (C.c_int * len(devices))(*devices) if devices else (C.c_int * 1)(device)
        r = abi.host().mrth_hip_render_devices(self._h, width, height, spp_x, spp_y, max_depth, seed, devs, len(devs), _fp(img),
                                               C.byref(st), C.byref(sec))
        if r < 0:
            raise MrtError(abi.host().mrth_last_error().decode())
        return img, st, sec.value


    def hip_render_progressive(self, width, height, spp_x, spp_y, samples_per_step, max_depth=5, seed=1, device=0,
                               checkpoint_after=-1, checkpoint_path=None):
        """HIPRenderer::Render in steps (the viewer's accumulation loop over sample ranges); with
        checkpoint_path the renderer is checkpointed after `checkpoint_after` steps, destroyed, and
        a fresh one resumes from the file.  Returns the radiance sums, statistics, step count."""
        img = np.zeros((height, width, 3), np.float32)
        st = abi.RenderStats()
        r = abi.host().mrth_hip_render_progressive(self._h, width, height, spp_x, spp_y, max_depth, seed, device, samples_per_step,
                                                   checkpoint_after, str(checkpoint_path).encode() if checkpoint_path else None,
                                                   _fp(img), C.byref(st))
        if r < 0:
            raise MrtError(abi.host().mrth_last_error().decode())
        return img, st, r


class KdTree:
    def __init__(self, scene, n_threads=0):
        self.scene = scene
        self._h = abi.host().mrth_build_kdtree(scene._h, n_threads)
        if not self._h:
            raise MrtError(abi.host().mrth_last_error().decode())

    def __del__(self):
        if getattr(self, "_h", None) and abi is not None and getattr(abi, "host", None):  # not during interpreter shutdown
            abi.host().mrth_tree_free(self._h)
            self._h = None

    def counts(self):
        a, b, d = C.c_int64(), C.c_int64(), C.c_int32()
        abi.host().mrth_tree_counts(self._h, C.byref(a), C.byref(b), C.byref(d))
        return a.value, b.value, d.value

    def dump(self):
        """Pre-order dump of the pointer tree (dict of numpy arrays)."""
        nn, nr, _ = self.counts()
        return tree_dump(lambda *ptrs: abi.host().mrth_tree_dump(self._h, *ptrs), nn, nr)

    def flatten(self, layout=1):
        return FlatScene(self, layout)

    def intersect(self, origin, direction, tmin=0.0, tmax=1e5, shadow=False):
        """intersectKdTree for ONE ray through the C++ host API (a GPU round trip per call: for
        debugging; batches go through DeviceScene.intersect).  Returns (prim or -1, t, u, v)."""
        o, d, out = _f32(origin, 3), _f32(direction, 3), np.zeros(4, np.float32)
        r = abi.host().mrth_intersect_kdtree(self.scene._h, self._h, _fp(o), _fp(d), tmin, tmax, 1 if shadow else 0, _fp(out))
        if r < 0:
            raise MrtError(abi.host().mrth_last_error().decode())
        return int(out[0]), float(out[1]), float(out[2]), float(out[3])


def tree_dump(call, nn, nr):
    out = {
        "leaf": np.zeros(nn, np.uint8), "axis": np.zeros(nn, np.int32), "split": np.zeros(nn, np.float32),
        "left": np.zeros(nn, np.int32), "right": np.zeros(nn, np.int32), "bmin": np.zeros((nn, 3), np.float32),
        "bmax": np.zeros((nn, 3), np.float32), "ref_off": np.zeros(nn, np.int32), "ref_cnt": np.zeros(nn, np.int32),
        "refs": np.zeros(max(nr, 1), np.int32),
    }
    order = ["leaf", "axis", "split", "left", "right", "bmin", "bmax", "ref_off", "ref_cnt", "refs"]
    call(*[out[k].ctypes.data_as(C.c_void_p) for k in order])
    out["refs"] = out["refs"][:nr]
    return out


class FlatScene:
    """flatten(scene, tree): owns the arrays behind an mrt_scene_desc."""

    def __init__(self, tree, layout=1):
        self.tree = tree
        self._h = abi.host().mrth_flatten(tree.scene._h, tree._h, layout)
        if not self._h:
            raise MrtError(abi.host().mrth_last_error().decode())
        self.desc = abi.host().mrth_flat_desc(self._h)  # POINTER(SceneDesc)
        self.depth = abi.host().mrth_flat_depth(self._h)

    def __del__(self):
        if getattr(self, "_h", None) and abi is not None and getattr(abi, "host", None):  # not during interpreter shutdown
            abi.host().mrth_flat_free(self._h)
            self._h = None

    def nodes(self):
        d = self.desc.contents
        n = d.n_nodes
        arr = np.ctypeslib.as_array(C.cast(d.nodes, abi.u32p), shape=(n * 2,)).reshape(n, 2).copy() if n else np.zeros((0, 2), np.uint32)
        refs = np.ctypeslib.as_array(d.refs, shape=(d.n_refs,)).copy() if d.n_refs else np.zeros(0, np.uint32)
        return arr, refs


def _is_torch(x):
    return type(x).__module__.startswith("torch")


class DeviceScene:
    """mrt_scene_create / mrt_scene_destroy on one device."""

    def __init__(self, flat, device=0):
        self.flat = flat
        self.device = device
        h = C.c_void_p()
        st = abi.hip().mrt_scene_create(flat.desc, device, C.byref(h))
        if st != abi.MRT_OK:
            raise MrtError("mrt_scene_create: %d %s" % (st, abi.hip().mrt_last_error().decode()))
        self._h = h

    def close(self):
        if getattr(self, "_h", None) and abi is not None and getattr(abi, "hip", None):  # not during interpreter shutdown
            abi.hip().mrt_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def intersect(self, o, d, tmin, tmax, any_hit=False, stats=False, stream=None):
        """Batched intersectKdTree.  o, d: [n,3]; tmin, tmax: [n].

        numpy inputs go through host buffers (MRT_BUF_HOST); torch CUDA tensors are used in place
        (MRT_BUF_DEVICE) and the outputs are torch tensors on the same device.
        """
        lib = abi.hip()
        flags = abi.MRT_ANY_HIT if any_hit else abi.MRT_CLOSEST_HIT
        ts = abi.TraceStats() if stats else None
        if _is_torch(o):
            import torch
            n = o.shape[0]
            cols = [o[:, 0].contiguous(), o[:, 1].contiguous(), o[:, 2].contiguous(), d[:, 0].contiguous(),
                    d[:, 1].contiguous(), d[:, 2].contiguous(), tmin.contiguous(), tmax.contiguous()]
            prim = torch.empty(n, dtype=torch.int32, device=o.device)
            t = torch.empty(n, dtype=torch.float32, device=o.device)
            u = torch.empty_like(t)
            v = torch.empty_like(t)
            rays = abi.Rays(*[c.data_ptr() for c in cols])
            hits = abi.Hits(prim.data_ptr(), t.data_ptr(), u.data_ptr(), v.data_ptr())
            s = stream if stream is not None else torch.cuda.current_stream(o.device).cuda_stream
            st = lib.mrt_intersect(self._h, n, C.byref(rays), C.byref(hits), flags, abi.MRT_BUF_DEVICE, s,
                                   C.byref(ts) if ts else None)
            if st != abi.MRT_OK:
                raise MrtError("mrt_intersect: %d %s" % (st, lib.mrt_last_error().decode()))
            return (prim, t, u, v, ts) if stats else (prim, t, u, v)
        o = np.asarray(o, np.float32)
        d = np.asarray(d, np.float32)
        n = o.shape[0]
        cols = [np.ascontiguousarray(o[:, k]) for k in range(3)] + [np.ascontiguousarray(d[:, k]) for k in range(3)] + \
               [np.ascontiguousarray(tmin, np.float32), np.ascontiguousarray(tmax, np.float32)]
        prim = np.empty(n, np.int32)
        t = np.empty(n, np.float32)
        u = np.empty(n, np.float32)
        v = np.empty(n, np.float32)
        rays = abi.Rays(*[c.ctypes.data for c in cols])
        hits = abi.Hits(prim.ctypes.data, t.ctypes.data, u.ctypes.data, v.ctypes.data)
        st = lib.mrt_intersect(self._h, n, C.byref(rays), C.byref(hits), flags, abi.MRT_BUF_HOST, None,
                               C.byref(ts) if ts else None)
        if st != abi.MRT_OK:
            raise MrtError("mrt_intersect: %d %s" % (st, lib.mrt_last_error().decode()))
        return (prim, t, u, v, ts) if stats else (prim, t, u, v)

    def trim(self):
        """mrt_scene_trim: gives back the workspace the handle keeps between calls."""
        if abi.hip().mrt_scene_trim(self._h) != abi.MRT_OK:
            raise MrtError(abi.hip().mrt_last_error().decode())

    def camera_choice(self):
        """mrt_debug_camera_choice: (packets 0 / 1, ms streaming, ms packets) of the handle's camera rays."""
        k, a, b = C.c_int32(), C.c_float(), C.c_float()
        if abi.hip().mrt_debug_camera_choice(self._h, C.byref(k), C.byref(a), C.byref(b)) != abi.MRT_OK:
            raise MrtError(abi.hip().mrt_last_error().decode())
        return k.value, a.value, b.value

    def relayed(self):
        """mrt_debug_relayed: rays this handle's traversal launches have handed over so far."""
        n = C.c_uint64()
        if abi.hip().mrt_debug_relayed(self._h, C.byref(n)) != abi.MRT_OK:
            raise MrtError(abi.hip().mrt_last_error().decode())
        return n.value

    def debug_eval(self, fn, rows, out_width):
        """mrt_debug_eval: one device shading function on rows of floats (diagnostics)."""
        rows = np.ascontiguousarray(rows, np.float32)
        n, w = rows.shape
        out = np.zeros((n, out_width), np.float32)
        st = abi.hip().mrt_debug_eval(self._h, fn, n, rows.ctypes.data_as(abi.f32p), w, out.ctypes.data_as(abi.f32p), out_width)
        if st != abi.MRT_OK:
            raise MrtError("mrt_debug_eval: %d %s" % (st, abi.hip().mrt_last_error().decode()))
        return out

    def render(self, params, rgb_sum, stream=None):
        """mrt_render: adds linear radiance sums into rgb_sum ([h,w,3] numpy array or torch CUDA tensor)."""
        lib = abi.hip()
        rs = abi.RenderStats()
        if _is_torch(rgb_sum):
            import torch
            s = stream if stream is not None else torch.cuda.current_stream(rgb_sum.device).cuda_stream
            st = lib.mrt_render(self._h, C.byref(params), rgb_sum.data_ptr(), abi.MRT_BUF_DEVICE, s, C.byref(rs))
        else:
            st = lib.mrt_render(self._h, C.byref(params), rgb_sum.ctypes.data, abi.MRT_BUF_HOST, None, C.byref(rs))
        if st != abi.MRT_OK:
            raise MrtError("mrt_render: %d %s" % (st, lib.mrt_last_error().decode()))
        return rs


def render_params(width, height, spp_x, spp_y, max_depth=5, seed=1, sample_begin=0, sample_end=None, row_begin=0,
                  row_end=None, paths_per_pass=0, flags=0, tile_step=0, tile_phase=0):
    p = abi.RenderParams()
    p.width, p.height, p.spp_x, p.spp_y = width, height, spp_x, spp_y
    p.sample_begin = sample_begin
    p.sample_end = spp_x * spp_y if sample_end is None else sample_end
    p.max_depth, p.seed = max_depth, seed
    p.row_begin = row_begin
    p.row_end = height if row_end is None else row_end
    p.paths_per_pass = paths_per_pass
    p.flags = flags
    p.tile_step, p.tile_phase = tile_step, tile_phase
    return p
