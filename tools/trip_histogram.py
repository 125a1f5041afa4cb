"""Trips of the traversal loop per ray, per bounce (needs a library built with -DMRT_TRIP_HISTOGRAM:
tools/build_variant.sh trips -DMRT_TRIP_HISTOGRAM, copied over raytracer_amd/libmrt_hip.so).
Renders the frame with max_depth 0, 1, ... and prints the histogram each further bounce adds.
usage: trip_histogram.py [scene] [n_samples]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from raytracer_amd import abi, scenes  # noqa: E402
from raytracer_amd.host import DeviceScene, render_params  # noqa: E402
W, H = 1920, 1080
name = sys.argv[1] if len(sys.argv) > 1 else "million_bunny"
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 128
sc = getattr(scenes, name)(W, H)
dev = DeviceScene(sc.initKdTree(16).flatten(1))
img = torch.zeros((H, W, 3), device='cuda')
lib = abi.hip()
buf = (C.c_ulonglong * 32)()
prev = [0] * 32
for d in range(0, 6):
    lib.mrt_debug_trip_histogram(buf, 1)
    p = render_params(W, H, 32, 32, max_depth=d, seed=1, sample_begin=0, sample_end=ns, flags=abi.MRT_RENDER_SERIAL)
    rs = dev.render(p, img)
    lib.mrt_debug_trip_histogram(buf, 1)
    cur = list(buf)
    dl = [a - b for a, b in zip(cur, prev)]
    tot = sum(dl)
    top = max(k for k in range(32) if cur[k])
    print("depth %d: +%d rays; trips 2^k..: %s" % (d, tot, " ".join("%d:%d" % (k, dl[k]) for k in range(32) if dl[k])), flush=True)
    prev = cur
