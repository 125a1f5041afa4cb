"""Wall time of every stage from files on disk to a finished frame, for one of the scene factories.
usage: pipeline_time.py [scene [spp_x spp_y]]"""
import sys
import time
sys.path.insert(0, '/root/repo')
import numpy as np
import torch
from raytracer_amd import abi, scenes
from raytracer_amd.host import DeviceScene, render_params

name = sys.argv[1] if len(sys.argv) > 1 else "million_bunny"
sx, sy = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (32, 32)
W, H = 1920, 1080
torch.zeros(1, device="cuda")
t = [time.perf_counter()]
sc = getattr(scenes, name)(W, H)
t.append(time.perf_counter())
tree = sc.initKdTree(16)
t.append(time.perf_counter())
flat = tree.flatten(1)
t.append(time.perf_counter())
dev = DeviceScene(flat)
t.append(time.perf_counter())
img = torch.zeros((H, W, 3), device="cuda")
p = render_params(W, H, sx, sy, max_depth=5, seed=1)
dev.render(p, img)
torch.cuda.synchronize()
t.append(time.perf_counter())
out = img.cpu().numpy()
t.append(time.perf_counter())
names = ["load + subdivide", "initKdTree (16 threads)", "flatten", "mrt_scene_create (upload)", "mrt_render %d spp" % (sx * sy), "image to host"]
for n, a, b in zip(names, t[:-1], t[1:]):
    print("%-28s %.3f s" % (n, b - a))
print("%-28s %.3f s" % ("total", t[-1] - t[0]), "mean radiance %.4f" % (out.mean() / (sx * sy)))
dev.close()
