"""Frame rate on a mesh whose objects come in random order, with the library's own object order
(default) and with the caller's order kept (MRT_OBJECT_ORDER=input).  usage: object_order_bench.py [scene] [spp_x spp_y]"""
import os
import sys
import time
sys.path.insert(0, '/root/repo')
import torch
from raytracer_amd import abi, scenes
from raytracer_amd.host import DeviceScene, render_params

name = sys.argv[1] if len(sys.argv) > 1 else "million_bunny"
sx, sy = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (16, 16)
W, H = 1920, 1080
img = torch.zeros((H, W, 3), device="cuda")
for shuffled in (False, True):
    sc = getattr(scenes, name)(W, H)
    if shuffled:
        abi.host().mrth_shuffle_objects(sc._h, 12345)
    flat = sc.initKdTree(16).flatten(1)
    for order in ("library", "input"):
        if order == "input":
            os.environ["MRT_OBJECT_ORDER"] = "input"
        else:
            os.environ.pop("MRT_OBJECT_ORDER", None)
        dev = DeviceScene(flat)
        p = render_params(W, H, sx, sy, max_depth=5, seed=1)
        dev.render(p, img)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rs = dev.render(p, img)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("%s mesh order %-8s object order %-8s %.1f ms %.0f Mrays/s" % (name, "shuffled" if shuffled else "file", order, dt * 1e3, (rs.rays_closest + rs.rays_any) / dt / 1e6), flush=True)
        dev.close()
