"""What one GPU of an N-way sample split of the bench frame costs: renders sample ranges [0, 1024/N) of
the 32 x 32 grid of the million_bunny frame (1920x1080, depth 5) and prints the time of each and
efficiency(N) = t(1024 spp) / (N * t(1024/N spp)).

usage: python tools/strong_proxy.py [--scene million_bunny] [--steps 3] [--only SPP] [--serial] [--paths-per-pass P]
  --only SPP   render just that range (one warm-up + --steps frames): for kernel traces
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from raytracer_amd import abi, scenes  # noqa: E402
from raytracer_amd.host import DeviceScene, render_params  # noqa: E402

W, H = 1920, 1080


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="million_bunny")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--only", type=int, default=0)
    ap.add_argument("--serial", action="store_true")
    ap.add_argument("--paths-per-pass", type=int, default=0)
    ap.add_argument("--spp-x", type=int, default=32)
    ap.add_argument("--spp-y", type=int, default=32)
    a = ap.parse_args()
    sc = getattr(scenes, a.scene)(W, H)
    tree = sc.initKdTree(16)
    dev = DeviceScene(tree.flatten(1))
    img = torch.zeros((H, W, 3), device="cuda")
    spp = a.spp_x * a.spp_y
    ranges = [a.only] if a.only else [spp // 8, spp // 4, spp // 2, spp]
    t = {}
    for k in ranges:
        p = render_params(W, H, a.spp_x, a.spp_y, max_depth=5, seed=1, sample_begin=0, sample_end=k,
                          paths_per_pass=a.paths_per_pass, flags=abi.MRT_RENDER_SERIAL if a.serial else 0)
        dev.render(p, img)
        torch.cuda.synchronize()
        best = 1e30
        for _ in range(a.steps):
            img.zero_()
            t0 = time.perf_counter()
            rs = dev.render(p, img)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        t[k] = best
        rays = rs.rays_closest + rs.rays_any
        print("%4d spp: %8.2f ms  %7.1f Mrays/s  passes %d  kernels ms: extend %.1f shade %.1f connect %.1f" % (
            k, best * 1e3, rays / best / 1e6, rs.passes, rs.extend_ms, rs.shade_ms, rs.connect_ms), flush=True)
    if not a.only:
        for k in ranges[:-1]:
            n = spp // k
            print("efficiency at %d GPUs (strong split, before the reduce): %.3f" % (n, t[spp] / (n * t[k])), flush=True)
    dev.close()


if __name__ == "__main__":
    main()
