"""Tree visits per ray by ray class (camera rays, rays leaving the first hit points towards random
directions of the hemisphere: closest-hit and any-hit), from mrt_intersect's counters.

usage: python tools/visit_stats.py [scene]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from raytracer_amd import scenes  # noqa: E402
from raytracer_amd.host import DeviceScene  # noqa: E402
from tests import orc  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "million_bunny"
W, H = 1920, 1080
sc = getattr(scenes, name)(W, H)
flat = sc.initKdTree(16).flatten(1)
dev = DeviceScene(flat)
o, d = orc.camera_rays(flat.desc, W, H, seed=1)
to, td = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()
g = torch.Generator(device="cuda")
g.manual_seed(7)


def run(label, oo, dd, any_hit):
    m = len(oo)
    prim, t, u, v, ts = dev.intersect(oo, dd, torch.zeros(m, device="cuda"), torch.full((m,), 1e5, device="cuda"), any_hit=any_hit, stats=True)
    print("%-34s %9d rays  hit %.2f  per ray: nodes %.1f  leaves %.2f  refs %.2f  tests %.2f" % (
        label, m, float((prim >= 0).float().mean()), ts.inner / m, ts.leaves / m, ts.refs / m, ts.prim_tests / m), flush=True)
    return prim, t


prim, t = run("camera", to, td, False)
hit = prim >= 0
p1 = (to + t[:, None] * td)[hit]
w = torch.randn((len(p1), 3), device="cuda", generator=g)
w = w / w.norm(dim=1, keepdim=True)
flip = (w * td[hit]).sum(1) > 0
w[flip] = -w[flip]
o1 = (p1 + 1e-3 * w).contiguous()
pr, t1 = run("from the first hits, closest", o1, w, False)
run("from the first hits, any", o1, w, True)
h1 = pr >= 0
o2 = (o1 + t1[:, None] * w)[h1]
w2 = torch.randn((len(o2), 3), device="cuda", generator=g)
w2 = w2 / w2.norm(dim=1, keepdim=True)
flip = (w2 * w[h1]).sum(1) > 0
w2[flip] = -w2[flip]
o2 = (o2 + 1e-3 * w2).contiguous()
run("from the second hits, closest", o2, w2, False)
run("from the second hits, any", o2, w2, True)
dev.close()
