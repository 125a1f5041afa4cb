"""Does the storage order of the triangles matter?  Renders the benchmark scene with the faces of
stanford-bunny.obj in file order, shuffled, and sorted by the Morton code of their centroids
(the object ids change with the order, so the images differ in tie-breaks only).

usage: python tools/prim_order_experiment.py [--spp-y 4]
"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from raytracer_amd import abi, scenes  # noqa: E402
from raytracer_amd.host import DeviceScene, Scene, Transform, render_params  # noqa: E402


def reorder(src, dst, mode, seed=1):
    lines = open(src).read().split("\n")
    faces = [i for i, l in enumerate(lines) if l.startswith("f ")]
    verts = np.array([[float(x) for x in l.split()[1:4]] for l in lines if l.startswith("v ")], np.float64)
    if mode == "shuffle":
        order = np.random.default_rng(seed).permutation(len(faces))
    else:
        idx = np.array([[int(t.split("/")[0]) - 1 for t in lines[i].split()[1:4]] for i in faces])
        c = verts[idx].mean(axis=1)
        q = ((c - c.min(0)) / (c.max(0) - c.min(0) + 1e-12) * 1023).astype(np.uint64)
        code = np.zeros(len(faces), np.uint64)
        for b in range(10):
            for k in range(3):
                code |= ((q[:, k] >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b + k)
        order = np.argsort(code, kind="stable")
    out = [l for i, l in enumerate(lines) if not l.startswith("f ")] + [lines[faces[j]] for j in order]
    open(dst, "w").write("\n".join(out) + "\n")


def scene(obj, w, h):
    sc = Scene()
    frm, at = (0.0, 2.8, 7.5), (0.0, 2.0, 0.0)
    focus = float(np.linalg.norm(np.array(at, np.float32) - np.array(frm, np.float32)).astype(np.float32))
    sc.setSimpleCamera(frm, at, (0, 1, 0), 60, w, h, 0.02, focus)
    sc.add_ibl_file(scenes.asset("brown_photostudio_06_1k.hdr"), also_env=True)
    floor = sc.CreateMaterial(abi.MRT_MAT_DIFFUSE, albedo=(0.6, 0.6, 0.6), metalness=0.0, roughness=0.8)
    sc.add_plane(Transform().scale(10, 10, 1).rotate((-1, 0, 0), 90).translate((0, 0.66, 0)), floor)
    diffuse = sc.CreateMaterial(abi.MRT_MAT_DIFFUSE, albedo=(0, 1, 0), roughness=0.5, metalness=0.1)
    glossy = sc.CreateMaterial(abi.MRT_MAT_SPECULAR, albedo=(1, 1, 1), roughness=0.2, metalness=0.9)
    sc.addObjectsFromFile(obj, diffuse, Transform().translate((-1.5, 0.0, 0)), prescale=20.0)
    sc.addObjectsFromFile(obj, glossy, Transform().translate((2.1, 0.0, 0)), prescale=20.0)
    return sc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spp-y", type=int, default=4)
    a = ap.parse_args()
    W, H = 1920, 1080
    tmp = tempfile.mkdtemp()
    src = scenes.asset("stanford-bunny.obj")
    files = {"file order": src}
    for mode in ("shuffle", "morton"):
        files[mode] = os.path.join(tmp, mode + ".obj")
        reorder(src, files[mode], mode)
    for name, path in files.items():
        sc = scene(path, W, H)
        dev = DeviceScene(sc.initKdTree().flatten(1))
        img = torch.zeros((H, W, 3), device="cuda")
        p = render_params(W, H, 16, a.spp_y, max_depth=5, seed=1)
        dev.render(p, img)
        torch.cuda.synchronize()
        best = None
        for _ in range(2):
            img.zero_()
            t0 = time.perf_counter()
            rs = dev.render(p, img)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        rays = rs.rays_closest + rs.rays_any
        print("%-10s %.1f ms  %.1f Mrays/s  extend %.1f connect %.1f shade %.1f" % (name, best * 1e3, rays / best / 1e6, rs.extend_ms, rs.connect_ms, rs.shade_ms), flush=True)
        dev.close()


if __name__ == "__main__":
    main()
