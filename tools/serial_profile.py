"""Per-class kernel times of one frame with one kernel at a time (MRT_RENDER_SERIAL) next to the
frame time with the default side-by-side execution.  usage: serial_profile.py scene [spp_x spp_y] [w h]"""
import sys
import time
sys.path.insert(0, '/root/repo')
import torch
from raytracer_amd import abi, scenes
from raytracer_amd.host import DeviceScene, render_params

name = sys.argv[1]
sx, sy = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (16, 16)
W, H = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (1920, 1080)
sc = scenes.parse_scene(int(name[5:]), W, H) if name.startswith("scene") else getattr(scenes, name)(W, H)
tree = sc.initKdTree()
dev = DeviceScene(tree.flatten(1))
img = torch.zeros((H, W, 3), device="cuda")
for label, flags in (("serial", abi.MRT_RENDER_SERIAL), ("overlapped", 0)):
    p = render_params(W, H, sx, sy, max_depth=5, seed=1, flags=flags)
    dev.render(p, img)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rs = dev.render(p, img)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rays = rs.rays_closest + rs.rays_any
    print("%s %-10s %.1f ms %.0f Mrays/s | extend %.1f shade %.1f shadow %.1f mis %.1f | rays: closest %.0fM (mis %.0fM) any %.0fM" % (
        name, label, dt * 1e3, rays / dt / 1e6, rs.extend_ms, rs.shade_ms, rs.connect_ms - rs.connect_mis_ms, rs.connect_mis_ms,
        rs.rays_closest / 1e6, rs.rays_mis / 1e6, rs.rays_any / 1e6), flush=True)
dev.close()
