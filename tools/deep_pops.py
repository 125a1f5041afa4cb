"""How often a wave takes a deferred node back from the scratch part of a lane's stack (needs a library built
with -DMRT_TRIP_HISTOGRAM -DMRT_COUNT_DEEP_POPS copied over raytracer_amd/libmrt_hip.so).  usage: python tools/deep_pops.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import abi, scenes
from raytracer_amd.host import DeviceScene, render_params
W, H = 1920, 1080
sc = scenes.million_bunny(W, H)
dev = DeviceScene(sc.initKdTree(16).flatten(1))
img = torch.zeros((H, W, 3), device='cuda')
lib = abi.hip(); buf = (C.c_ulonglong * 32)()
for flags, label in ((abi.MRT_RENDER_SERIAL, "serial"),):
    lib.mrt_debug_trip_histogram(buf, 1)
    p = render_params(W, H, 32, 32, max_depth=5, seed=1, sample_begin=0, sample_end=64, flags=flags)
    dev.render(p, img); torch.cuda.synchronize()
    lib.mrt_debug_trip_histogram(buf, 1)
    b = list(buf)
    print(label, "wave-trips", b[28], "with a scratch pop %.3f" % (b[29] / b[28]), "with a lane whose stack is in scratch %.3f" % (b[30] / b[28]))
