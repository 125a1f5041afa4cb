"""One frame rendered through HIPRenderer with one handle and with two handles on the SAME device
(mrt_render_multi: the samples divided between two concurrent pipelines).  usage: two_handles_bench.py [scene] [spp_x spp_y]"""
import sys
import time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import scenes

name = sys.argv[1] if len(sys.argv) > 1 else "million_bunny"
sx, sy = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (32, 32)
W, H = 1920, 1080
sc = getattr(scenes, name)(W, H)
for devices in ([0], [0, 0], [0], [0, 0], [0, 0, 0]):
    img, rs, sec = sc.hip_render(W, H, sx, sy, max_depth=5, seed=1, devices=devices)
    print("%s handles %d: run() %.3f s, kernels extend %.0f shade %.0f connect %.0f ms, %.0f Mrays" % (
        name, len(devices), sec, rs.extend_ms, rs.shade_ms, rs.connect_ms, (rs.rays_closest + rs.rays_any) / 1e6), flush=True)
