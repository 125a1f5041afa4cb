"""Wall time of initKdTree for one scene factory, with and without PyTorch and a HIP context in the process.
usage: build_time.py [scene] [torch]"""
import sys
import time
sys.path.insert(0, '/root/repo')
if len(sys.argv) > 2 and sys.argv[2] == "torch":
    import torch
    torch.zeros(1, device="cuda")
from raytracer_amd import scenes
sc = getattr(scenes, sys.argv[1] if len(sys.argv) > 1 else "million_bunny")(1920, 1080)
for rep in range(2):
    t0 = time.perf_counter()
    tree = sc.initKdTree(16)
    print("initKdTree %.3f s" % (time.perf_counter() - t0), sys.argv[2:] or "no torch", flush=True)
    del tree
