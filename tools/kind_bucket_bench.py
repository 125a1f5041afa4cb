"""Shade time with and without regrouping the paths of a block by material kind, on a scene with
all three kinds (scenes.mixed_kinds) and on one with a single kind (scenes.bunny_pair, where the
regrouping is never switched on: reported for reference).  usage: kind_bucket_bench.py [spp_y]"""
import sys
sys.path.insert(0, '/root/repo')
import torch
from raytracer_amd import abi, scenes
from raytracer_amd.host import DeviceScene, render_params

W, H = 1920, 1080
spp_y = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for name in ("mixed_kinds", "bunny_pair"):
    sc = getattr(scenes, name)(W, H)
    tree = sc.initKdTree()
    flat = tree.flatten(1)
    dev = DeviceScene(flat)
    img = torch.zeros((H, W, 3), device="cuda")
    for label, flags in (("bucketed", 0), ("plain", abi.MRT_RENDER_NO_KIND_BUCKETS), ("bucketed", 0), ("plain", abi.MRT_RENDER_NO_KIND_BUCKETS)):
        for depth in (5, 8):
            p = render_params(W, H, 16, spp_y, max_depth=depth, seed=1, flags=flags | abi.MRT_RENDER_SERIAL)
            dev.render(p, img)
            rs = dev.render(p, img)
            torch.cuda.synchronize()
            print("%-12s %-9s depth %d: shade %.1f ms  extend %.1f  connect %.1f  rays %.1fM" % (
                name, label, depth, rs.shade_ms, rs.extend_ms, rs.connect_ms, (rs.rays_closest + rs.rays_any) / 1e6), flush=True)
    dev.close()
