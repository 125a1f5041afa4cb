#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ from the REFERENCE's own code.

Runs only in the build container: it needs oracle/_ref/libmrt_ref.so (the reference's kdtree / shape
/ sampling / bsdf / camera translation units compiled by oracle/Makefile) and /root/reference for
src/expected.h.  The fixtures are data -- seeded inputs and the reference's outputs -- and are what
pins the CPU oracle on machines where the reference does not exist.

    python tests/golden/make_golden.py
"""
import ctypes as C
import json
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from raytracer_amd import scenes  # noqa: E402
from tests import orc  # noqa: E402
from tests.golden import cases  # noqa: E402

P = orc._p


def expected_h():
    """src/expected.h: known-answer tables for RDM_Beckmann and RDM_Fresnel (values only)."""
    txt = open("/root/reference/src/expected.h").read()
    rows = [[float(x) for x in m.group(1).split(",")] for m in re.finditer(r"\{([-0-9.eE+, ]+)\}", txt)]
    beck = [r for r in rows if len(r) == 3]
    fres = [r for r in rows if len(r) == 4]
    with open(os.path.join(HERE, "expected_h.json"), "w") as f:
        json.dump({"source": "src/expected.h", "beckmann_ndoth_alpha_res": beck, "fresnel_ldoth_ext_int_res": fres}, f, indent=0)
    print("expected.h: %d Beckmann rows, %d Fresnel rows" % (len(beck), len(fres)))


def transforms():
    out = {}
    for name, ops in cases.TRANSFORM_CASES.items():
        m, inv, t = orc.ref_transform(ops)
        out[name] = {"ops": [list(o) for o in ops], "m": m.view(np.uint32).tolist(), "inv": inv.view(np.uint32).tolist(),
                     "t": t.view(np.uint32).tolist()}
    with open(os.path.join(HERE, "transforms.json"), "w") as f:
        json.dump(out, f)
    print("transforms: %d cases" % len(out))


def trees_and_hits():
    lib = orc.ref()
    summary = {}
    for name in cases.SCENES:
        sc, w, h = cases.make_scene(name)
        tree = sc.initKdTree()
        flat = tree.flatten(0)
        rs = orc.RefScene(flat.desc, orc.xf_ops_for(flat, sc))
        d = rs.dump()
        canon = orc.canonical_from_dump(d)
        summary[name] = {"n_objects": sc.n_objects, "n_nodes": int(len(d["leaf"])), "n_refs": int(len(d["refs"])),
                         "depth": int(d["depth"]), "digest": orc.tree_digest(canon),
                         "root_min": d["bmin"][0].view(np.uint32).tolist(), "root_max": d["bmax"][0].view(np.uint32).tolist()}
        arrays = {}
        if len(d["leaf"]) < 2000:  # the whole tree for small scenes
            arrays.update(tree_axis=canon[0], tree_split=canon[1], tree_counts=canon[2], tree_refs=canon[3])
        o, dd, tmin, tmax = cases.rays_for(flat, w, h, name)
        arrays["ray_digest"] = np.frombuffer(cases.digest(o, dd, tmin, tmax).encode(), np.uint8)
        for shadow in (False, True):
            r = rs.intersect(o, dd, tmin, tmax, shadow)
            k = "any" if shadow else "closest"
            arrays[k + "_hit"] = r["hit"]
            arrays[k + "_prim"] = r["prim"]
            arrays[k + "_t"] = r["t"].view(np.uint32)
            if not shadow:
                arrays[k + "_u"] = r["u"].view(np.uint32)
                arrays[k + "_v"] = r["v"].view(np.uint32)
                arrays[k + "_nrm"] = r["nrm"].view(np.uint32)
                arrays[k + "_outside"] = r["outside"]
        np.savez_compressed(os.path.join(HERE, "scene_%s.npz" % name), **arrays)
        print("scene %s: %d nodes, %d refs, depth %d, %d rays, %d closest hits" % (
            name, len(d["leaf"]), len(d["refs"]), d["depth"], len(o), int(arrays["closest_hit"].sum())))
    with open(os.path.join(HERE, "trees.json"), "w") as f:
        json.dump(summary, f, indent=1)


def large_trees():
    """Digests of the reference's kd-trees for the two benchmark scenes (138 902 and 1 111 216
    triangles).  The reference build of the larger one takes minutes and several GB."""
    import time
    summary = {}
    for name in ("bunny_pair", "million_bunny"):
        sc = getattr(scenes, name)(64, 36)
        tree = sc.initKdTree()
        flat = tree.flatten(0)
        t0 = time.time()
        rs = orc.RefScene(flat.desc, orc.xf_ops_for(flat, sc))
        d = rs.dump()
        canon = orc.canonical_from_dump(d)
        summary[name] = {"n_objects": sc.n_objects, "n_nodes": int(len(d["leaf"])), "n_refs": int(len(d["refs"])),
                         "depth": int(d["depth"]), "digest": orc.tree_digest(canon),
                         "root_min": d["bmin"][0].view(np.uint32).tolist(), "root_max": d["bmax"][0].view(np.uint32).tolist(),
                         "reference_build_seconds": round(time.time() - t0, 1)}
        print("large tree %s: %d nodes, %d refs, depth %d (%.0f s)" % (name, len(d["leaf"]), len(d["refs"]), d["depth"], time.time() - t0))
        del rs, d, canon
    with open(os.path.join(HERE, "trees_large.json"), "w") as f:
        json.dump(summary, f, indent=1)


def functions():
    lib = orc.ref()
    g = cases.function_inputs()
    out = {}
    n = g["n"]
    f32 = lambda *shape: np.zeros(shape, np.float32)
    # bsdf.cpp
    a = f32(n); lib.ref_beckmann(n, P(g["ndoth"]), P(g["alpha"]), P(a)); out["beckmann"] = a
    a = f32(n); lib.ref_fresnel(n, P(g["ldoth_signed"]), P(g["ext"]), P(g["int"]), P(a)); out["fresnel"] = a
    a = f32(n); lib.ref_smith(n, P(g["ldoth"]), P(g["ldotn"]), P(g["vdoth"]), P(g["vdotn"]), P(g["alpha"]), P(a)); out["smith"] = a
    a = f32(n, 3); lib.ref_bsdf(n, P(g["bsdf_in"]), P(a)); out["bsdf"] = a
    f0, df, F = f32(n, 3), f32(n, 3), f32(n, 3)
    lib.ref_schlick_f(n, P(g["base"]), P(g["metal"]), P(g["ldoth"]), P(f0), P(df), P(F))
    out["schlick_f0"], out["schlick_diffuse"], out["schlick_F"] = f0, df, F
    # warps / heuristics / distributions
    a = f32(n, 2); lib.ref_uniform_sample_triangle(n, P(g["u2"]), P(a)); out["uniform_sample_triangle"] = a
    a = f32(n, 2); lib.ref_concentric_disk(n, P(g["u2"]), P(a)); out["concentric_disk"] = a
    a = f32(n); lib.ref_power_heuristic(n, P(g["pdf_f"]), P(g["pdf_g"]), P(a)); out["power_heuristic"] = a
    for nl in cases.LIGHT_COUNTS:
        func = np.ones(nl, np.float32)
        off, pdf = np.zeros(n, np.int32), f32(n)
        lib.ref_dist1d_discrete(P(func), nl, n, P(g["u1"]), P(off), P(pdf))
        out["dist1d_off_%d" % nl], out["dist1d_pdf_%d" % nl] = off, pdf
    env = scenes.synthetic_env(*cases.ENV_SIZE)
    a = f32(n); lib.ref_dist2d_pdf(P(env.reshape(-1)), cases.ENV_SIZE[0], cases.ENV_SIZE[1], n, P(g["u2"]), P(a)); out["dist2d_pdf"] = a
    # rng / sampler
    a = np.zeros(256, np.uint32); lib.ref_rng_u32(cases.RNG_SEQ, 256, P(a)); out["rng_u32"] = a
    a = f32(256); lib.ref_rng_float(cases.RNG_SEQ, 256, P(a)); out["rng_float"] = a
    a = np.zeros(256, np.uint32); lib.ref_rng_bounded(cases.RNG_SEQ, 256, P(g["bounds"]), P(a)); out["rng_bounded"] = a
    for (nx, ny, seed, npix, nget) in cases.STRATIFIED_CASES:
        a = f32(npix * nx * ny * (5 + 7 * nget))
        lib.ref_stratified_stream(nx, ny, 8, seed, npix, nget, P(a))
        out["stratified_%dx%d_%d" % (nx, ny, seed)] = a
    # camera
    for name, c in cases.CAMERA_CASES.items():
        frm, at, up = [np.asarray(x, np.float32) for x in c[:3]]
        cam = lib.ref_camera_new(P(frm), P(at), P(up), *[float(x) for x in c[3:]])
        par = f32(12); lib.ref_camera_params(cam, P(par)); out["camera_%s_params" % name] = par
        o, d = f32(n, 3), f32(n, 3)
        lib.ref_camera_rays(cam, n, P(g["st"]), P(g["u2"]), P(o), P(d))
        out["camera_%s_o" % name], out["camera_%s_d" % name] = o, d
    # aabb
    hit, t0, t1 = np.zeros(n, np.uint8), f32(n), f32(n)
    lib.ref_intersect_aabb(n, P(g["ray_o"]), P(g["ray_d"]), P(g["tmin"]), P(g["tmax"]), P(g["box_min"]), P(g["box_max"]), P(hit), P(t0), P(t1))
    out["aabb_hit"], out["aabb_tmin"], out["aabb_tmax"] = hit, t0, t1
    # shapes: single-object intersect, sample, area, pdf through a scene holding one object each
    sc = cases.shape_scene()
    flat = sc.initKdTree().flatten(0)
    rs = orc.RefScene(flat.desc, orc.xf_ops_for(flat, sc))
    for obj, name in enumerate(cases.SHAPE_NAMES):
        r = orc.trace_out(n)
        sh = np.zeros(n, np.uint8)
        lib.ref_object_intersect(rs.scene, obj, n, P(g["ray_o"]), P(g["ray_d"]), P(g["tmin"]), P(g["tmax"]), P(sh), P(r["hit"]),
                                 P(r["t"]), P(r["u"]), P(r["v"]), P(r["pos"]), P(r["nrm"]), P(r["outside"]))
        for k in ("hit", "t", "u", "v", "pos", "nrm", "outside"):
            out["shape_%s_%s" % (name, k)] = r[k]
        if name != "plane":
            pos, nrm, pdf = f32(n, 3), f32(n, 3), f32(n)
            lib.ref_object_sample(rs.scene, obj, n, P(g["surf_pos"]), P(g["surf_nrm"]), P(g["outside"]), P(g["u2"]), P(pos), P(nrm), P(pdf))
            out["shape_%s_sample_pos" % name], out["shape_%s_sample_nrm" % name], out["shape_%s_sample_pdf" % name] = pos, nrm, pdf
            a = f32(n)
            lib.ref_object_pdf(rs.scene, obj, n, P(g["surf_pos"]), P(g["surf_nrm"]), P(g["toward_%s" % name]), P(a))
            out["shape_%s_pdf" % name] = a
            out["shape_%s_area" % name] = np.array([lib.ref_object_area(rs.scene, obj)], np.float32)
    np.savez_compressed(os.path.join(HERE, "functions.npz"), **{k: (v.view(np.uint32) if v.dtype == np.float32 else v) for k, v in out.items()})
    print("functions: %d tables of %d rows" % (len(out), n))


def shading_l2():
    """textures.hpp, random_dir, CoordinateSystem, UniformLightDistribution: the header-only / small
    pieces of the shading layer that build without Light.h."""
    lib = orc.ref()
    g = cases.shading_inputs()
    n = g["n"]
    out = {}
    h = orc.ref_transform_handle(cases.TEXTURE_XF_OPS)
    a = np.zeros((n, 3), np.float32)
    lib.ref_checker_value(P(np.asarray(cases.CHECKER[0], np.float32)), P(np.asarray(cases.CHECKER[1], np.float32)), h, n, P(g["uv"]), P(a))
    out["checker_value"] = a
    a = np.zeros((n, 3), np.float32)
    px = np.ascontiguousarray(cases.texture_pixels())
    if lib.ref_image_value(b"/tmp/_mrt_golden_texture.png", cases.TEXTURE_SIZE[0], cases.TEXTURE_SIZE[1], P(px), h, n, P(g["uv"]), P(a)) != 0:
        raise SystemExit("could not write the texture PNG")
    out["image_value"] = a
    draws, d = np.zeros((n, 5), np.float32), np.zeros((n, 3), np.float32)
    lib.ref_random_dir(n, P(g["normal"]), P(g["seed"]), P(draws), P(d))
    out["random_dir_draws"], out["random_dir"] = draws, d
    a, b = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
    lib.ref_coordinate_system(n, P(g["normal"]), P(a), P(b))
    out["coordinate_system_v2"], out["coordinate_system_v3"] = a, b
    for nl in cases.LIGHT_COUNTS:
        off, pdf = np.zeros(n, np.int32), np.zeros(n, np.float32)
        lib.ref_uniform_light_distribution(nl, n, P(g["u1"]), P(off), P(pdf))
        out["uniform_light_off_%d" % nl], out["uniform_light_pdf_%d" % nl] = off, pdf
    np.savez_compressed(os.path.join(HERE, "shading_l2.npz"), **{k: (v.view(np.uint32) if v.dtype == np.float32 else v) for k, v in out.items()})
    print("shading_l2: %d tables of %d rows" % (len(out), n))


def ref_obj_streams(path):
    """tinyobj::LoadObj as scene.cpp:179-187 calls it + the face walk of example_scenes.cpp:32-81."""
    lib = orc.ref()
    nf, nm, hn, ht = C.c_int64(), C.c_int64(), C.c_int32(), C.c_int32()
    mdir = (os.path.dirname(path) + "/").encode()
    if lib.ref_obj_load(path.encode(), mdir, C.byref(nf), C.byref(nm), C.byref(hn), C.byref(ht), None, None, None, None, None) != 0:
        raise SystemExit("tinyobj could not load " + path)
    n = nf.value
    pos, nrm, tex = np.zeros((n, 9), np.float32), np.zeros((n, 9), np.float32), np.zeros((n, 6), np.float32)
    mid, mtl = np.zeros(n, np.int32), np.zeros((max(nm.value, 1), 10), np.float32)
    lib.ref_obj_load(path.encode(), mdir, C.byref(nf), C.byref(nm), C.byref(hn), C.byref(ht), P(pos), P(nrm), P(tex), P(mid), P(mtl))
    face, vn = np.zeros((n, 3), np.float32), np.zeros((n, 9), np.float32)
    lib.ref_face_normals(n, P(pos), hn.value, P(nrm), P(face), P(vn))
    return {"pos": pos, "vertex_normals": vn, "face_normal": face, "tex": tex, "material_id": mid, "mtl": mtl[:nm.value],
            "has_normals": hn.value, "has_texcoords": ht.value}


def obj_streams():
    arrays, summary = {}, {}
    for name in cases.OBJ_FILES:
        s = ref_obj_streams(scenes.asset(name))
        key = name.replace("/", "_").replace(".obj", "").replace("-", "_")
        n = len(s["pos"])
        summary[name] = {"key": key, "faces": n, "materials": int(len(s["mtl"])), "has_normals": s["has_normals"],
                         "has_texcoords": s["has_texcoords"],
                         "digest": {k: cases.digest(s[k]) for k in ("pos", "vertex_normals", "tex", "material_id")}}
        arrays[key + "_mtl"] = s["mtl"].view(np.uint32)
        if n <= cases.OBJ_FULL_LIMIT:
            for k in ("pos", "vertex_normals", "face_normal", "tex"):
                arrays[key + "_" + k] = s[k].view(np.uint32)
            arrays[key + "_material_id"] = s["material_id"]
        print("obj %s: %d faces, %d MTL materials" % (name, n, len(s["mtl"])))
    np.savez_compressed(os.path.join(HERE, "obj_streams.npz"), **arrays)
    with open(os.path.join(HERE, "obj_streams.json"), "w") as f:
        json.dump(summary, f, indent=1)


def obj_fuzz():
    """Digests of the tinyobj streams of the first 100 files of tests/fuzz_obj.py, for machines without
    the reference build."""
    import tempfile
    from tests import fuzz_obj
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for seed in range(100):
            s = ref_obj_streams(fuzz_obj.make_obj(seed, tmp, "f%d" % seed))
            out[str(seed)] = {"faces": len(s["pos"]), "materials": int(len(s["mtl"])),
                              "digest": {k: cases.digest(s[k]) for k in ("pos", "vertex_normals", "tex", "material_id")},
                              "mtl": cases.digest(np.ascontiguousarray(s["mtl"]))}
    with open(os.path.join(HERE, "obj_fuzz.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("obj_fuzz: %d files" % len(out))


if __name__ == "__main__":
    if not orc.have_ref():
        raise SystemExit("oracle/_ref/libmrt_ref.so is missing: run `make -C oracle ref` in the build container")
    expected_h()
    transforms()
    trees_and_hits()
    functions()
    shading_l2()
    obj_streams()
    obj_fuzz()
    if "--large-trees" in sys.argv:
        large_trees()
