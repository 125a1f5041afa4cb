"""What ordering a list of incoherent rays by origin cell is worth: second-generation rays (leaving
the hit points of rays that left the primary hit points, as the continuation / shadow / MIS rays of
bounce >= 1 do), traced in the order of their parents (the order the render queues have) and
stable-sorted by the Morton code of their origin quantised to 2^bits cells per axis.

usage: python tools/sort_experiment.py [scene] [mult]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from raytracer_amd import scenes  # noqa: E402
from raytracer_amd.host import DeviceScene  # noqa: E402
from tests import orc  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "bunny_pair"
mult = int(sys.argv[2]) if len(sys.argv) > 2 else 16
W, H = 1920, 1080
sc = getattr(scenes, name)(W, H, env=None)
tree = sc.initKdTree()
flat = tree.flatten(1)
dev = DeviceScene(flat)
o, d = orc.camera_rays(flat.desc, W, H, seed=1)
idx = np.arange(W * H).reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(-1)  # 8x8 pixel tiles
to, td = torch.from_numpy(np.ascontiguousarray(o[idx])).cuda(), torch.from_numpy(np.ascontiguousarray(d[idx])).cuda()
n = len(to)
g = torch.Generator(device="cuda")
g.manual_seed(7)


def hemi(normal_like, m):
    w = torch.randn((m, 3), device="cuda", generator=g)
    w = w / w.norm(dim=1, keepdim=True)
    flip = (w * normal_like).sum(1) > 0
    w[flip] = -w[flip]
    return w


def trace(oo, dd, far=1e5, any_hit=False):
    m = len(oo)
    return dev.intersect(oo, dd, torch.zeros(m, device="cuda"), torch.full((m,), far, device="cuda"), any_hit=any_hit)


def timed(oo, dd, far, any_hit, reps=3):
    m = len(oo)
    tmin, tmax = torch.zeros(m, device="cuda"), torch.full((m,), far, device="cuda")
    dev.intersect(oo, dd, tmin, tmax, any_hit=any_hit)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        dev.intersect(oo, dd, tmin, tmax, any_hit=any_hit)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


prim, t = trace(to, td)[:2]
hit = prim >= 0
# first generation: `mult` rays per primary hit, the samples of a pixel together (the renderer's order)
p1 = (to + t[:, None] * td)[hit]
d0 = td[hit]
p1 = p1.repeat_interleave(mult, dim=0)
w1 = hemi(d0.repeat_interleave(mult, dim=0), len(p1))
o1 = (p1 + 1e-3 * w1).contiguous()
print("%s: first generation %d rays" % (name, len(o1)), flush=True)
pr, t1 = trace(o1, w1)[:2]
h1 = pr >= 0
o2p = (o1 + t1[:, None] * w1)[h1]
w2 = hemi(w1[h1], len(o2p))
o2 = (o2p + 1e-3 * w2).contiguous()
d2 = w2.contiguous()
print("second generation %d rays" % len(o2), flush=True)
lo, hi = o2.min(0).values, o2.max(0).values


def morton(bits):
    q = ((o2 - lo) / (hi - lo + 1e-9) * (2 ** bits - 1)).long().clamp(0, 2 ** bits - 1)
    code = torch.zeros(len(o2), dtype=torch.long, device="cuda")
    for b in range(bits):
        for k in range(3):
            code |= ((q[:, k] >> b) & 1) << (3 * b + k)
    return code


for label, far, any_hit in (("closest", 1e4, False), ("any-hit", 1e4, True)):
    base = timed(o2, d2, far, any_hit)
    print("%-8s parent order            %.2f ms  %.1f Mrays/s" % (label, base, len(o2) / base / 1e3), flush=True)
    for bits in (3, 4, 5, 6, 10):
        order = torch.sort(morton(bits), stable=True).indices
        ms = timed(o2[order].contiguous(), d2[order].contiguous(), far, any_hit)
        print("%-8s %2d bits/axis (%7d cells) %.2f ms  %.1f Mrays/s  (x%.2f)" % (label, bits, 8 ** bits, ms, len(o2) / ms / 1e3, base / ms), flush=True)
dev.close()
