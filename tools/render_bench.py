"""Whole-frame rate of mrt_render on one of the scene factories (not the bench.py contract: a tool
for the other BASELINE size classes).

usage: python tools/render_bench.py [--scene cornell|bunny_pair|million_bunny|veach|plant] [--spp-x 16 --spp-y 4] [--depth 5]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from raytracer_amd import scenes  # noqa: E402
from raytracer_amd.host import DeviceScene, render_params  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="million_bunny")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp-x", type=int, default=16)
    ap.add_argument("--spp-y", type=int, default=4)
    ap.add_argument("--depth", type=int, default=5)
    ap.add_argument("--steps", type=int, default=2)
    a = ap.parse_args()
    factory = {"cornell": scenes.cornell, "bunny_pair": scenes.bunny_pair, "million_bunny": scenes.million_bunny,
               "veach": scenes.veach_mi, "plant": scenes.plant}[a.scene]
    t0 = time.time()
    sc = factory(a.width, a.height)
    tree = sc.initKdTree()
    print("%s: %d objects, tree (nodes, refs, depth) %s, set-up %.1f s" % (a.scene, sc.n_objects, tree.counts(), time.time() - t0), flush=True)
    dev = DeviceScene(tree.flatten(1))
    img = torch.zeros((a.height, a.width, 3), device="cuda")
    p = render_params(a.width, a.height, a.spp_x, a.spp_y, max_depth=a.depth, seed=1)
    dev.render(p, img)
    torch.cuda.synchronize()
    for _ in range(a.steps):
        img.zero_()
        t0 = time.perf_counter()
        rs = dev.render(p, img)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        rays = rs.rays_closest + rs.rays_any
        print("%.1f ms  %.1f Mrays/s  %.2f rays/path  kernels ms: extend %.1f shade %.1f connect %.1f  mean radiance %.4f" % (
            dt * 1e3, rays / dt / 1e6, rays / rs.paths, rs.extend_ms, rs.shade_ms, rs.connect_ms,
            float(img.mean()) / (a.spp_x * a.spp_y)), flush=True)
    dev.close()


if __name__ == "__main__":
    main()
