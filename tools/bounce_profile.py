"""Rays and device time added by every bounce of a workload: renders the frame with max_depth
0, 1, ... and prints the differences.  usage: bounce_profile.py [scene] [spp_x] [spp_y]"""
import sys
sys.path.insert(0, '/root/repo')
import torch
from raytracer_amd import scenes
from raytracer_amd.host import DeviceScene, render_params
W, H = 1920, 1080
name = sys.argv[1] if len(sys.argv) > 1 else "bunny_pair"
sx = int(sys.argv[2]) if len(sys.argv) > 2 else 16
sy = int(sys.argv[3]) if len(sys.argv) > 3 else 16
sc = getattr(scenes, name)(W, H)
tree = sc.initKdTree()
flat = tree.flatten(1)
dev = DeviceScene(flat)
img = torch.zeros((H, W, 3), device='cuda')
prev = None
for d in range(0, 6):
    p = render_params(W, H, sx, sy, max_depth=d, seed=1)
    dev.render(p, img)
    rs = dev.render(p, img)
    cur = (rs.rays_closest, rs.rays_any, rs.rays_mis, rs.extend_ms, rs.connect_ms, rs.shade_ms, rs.connect_mis_ms)
    dl = cur if prev is None else tuple(a - b for a, b in zip(cur, prev))
    ext_rays = dl[0] - dl[2]
    print("depth %d: +extend %.1fM rays %.1f ms (%.2f G/s) | +shadow %.1fM %.1f ms (%.2f G/s) | +mis %.1fM %.1f ms (%.2f G/s) | +shade %.1f ms" % (
        d, ext_rays / 1e6, dl[3], ext_rays / max(dl[3], 1e-9) / 1e6, dl[1] / 1e6, dl[4] - dl[6], dl[1] / max(dl[4] - dl[6], 1e-9) / 1e6,
        dl[2] / 1e6, dl[6], dl[2] / max(dl[6], 1e-9) / 1e6, dl[5]), flush=True)
    prev = cur
