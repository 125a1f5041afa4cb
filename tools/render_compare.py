"""Renders small scenes on the GPU and with the CPU oracle (same sample streams) and prints
difference statistics.  usage: python tools/render_compare.py [cornell|bunny|veach|all]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from raytracer_amd import scenes  # noqa: E402
from raytracer_amd.host import DeviceScene, render_params  # noqa: E402
from tests import orc  # noqa: E402


def rel_mse(a, b):
    return float(np.mean((a - b) ** 2 / (b ** 2 + 1e-2)))


def compare(name, sc, w, h, sx, sy, depth=5, seed=7):
    tree = sc.initKdTree()
    flat = tree.flatten(1)
    dev = DeviceScene(flat)
    p = render_params(w, h, sx, sy, max_depth=depth, seed=seed)
    gpu = np.zeros((h, w, 3), np.float32)
    t0 = time.time()
    rs = dev.render(p, gpu)
    t1 = time.time()
    cpu, crs, cts, _ = orc.oracle_render(flat.desc, p, 0, threads=16)
    t2 = time.time()
    spp = sx * sy
    g, c = gpu / spp, cpu / spp
    diff = np.abs(g - c)
    rel = diff / (np.abs(c) + 1e-3)
    print("%-8s %dx%d %dspp: gpu %.2fs cpu %.2fs | rays gpu closest %d any %d | cpu closest %d any %d" % (
        name, w, h, spp, t1 - t0, t2 - t1, rs.rays_closest, rs.rays_any, crs.rays_closest, crs.rays_any))
    print("   mean gpu %.6f cpu %.6f  relMSE %.3e  max abs diff %.3e  pixels with rel diff >1e-3: %d/%d  >1e-5: %d" % (
        g.mean(), c.mean(), rel_mse(g, c), diff.max(), int((rel.max(axis=2) > 1e-3).sum()), w * h,
        int((rel.max(axis=2) > 1e-5).sum())))
    print("   kernel ms: generate %.2f extend %.2f shade %.2f connect %.2f" % (rs.generate_ms, rs.extend_ms, rs.shade_ms, rs.connect_ms))
    dev.close()
    return g, c


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("cornell", "all"):
        compare("cornell", scenes.cornell(96, 96), 96, 96, 4, 4)
    if which in ("bunny", "all"):
        compare("bunny", scenes.small_bunny(96, 96, env_pixels=scenes.synthetic_env()), 96, 96, 4, 4)
    if which in ("veach", "all"):
        compare("veach", scenes.veach_mi(128, 72), 128, 72, 4, 4)


if __name__ == "__main__":
    main()
