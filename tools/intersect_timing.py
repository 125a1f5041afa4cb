"""Times mrt_intersect on primary and incoherent rays (device-resident buffers).

usage: python tools/intersect_timing.py [scene] [--subdiv N]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from raytracer_amd import scenes  # noqa: E402
from raytracer_amd.host import DeviceScene  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("scene", nargs="?", default="bunny")
    ap.add_argument("--subdiv", type=int, default=0)
    ap.add_argument("--layout", type=int, default=1)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    W, H = 1920, 1080
    t0 = time.time()
    if a.scene == "bunny":
        sc = scenes.bunny_ibl(W, H, env=None, subdivisions=a.subdiv)
    elif a.scene == "cornell":
        sc = scenes.cornell(W, H)
    else:
        sc = scenes.veach_mi(W, H)
    tree = sc.initKdTree()
    print("objects %d, tree (nodes, refs, depth) %s, build %.1fs" % (sc.n_objects, tree.counts(), time.time() - t0), flush=True)
    flat = tree.flatten(a.layout)
    dev = DeviceScene(flat)

    from tests import orc
    o, d = orc.camera_rays(flat.desc, W, H, seed=1)
    o2, d2 = orc.random_rays(flat.desc, W * H, seed=2)
    for name, oo, dd in (("primary", o, d), ("incoherent", o2, d2)):
        n = len(oo)
        to = torch.from_numpy(np.ascontiguousarray(oo)).cuda()
        td = torch.from_numpy(np.ascontiguousarray(dd)).cuda()
        tmin = torch.zeros(n, device="cuda")
        tmax = torch.full((n,), 1e5, device="cuda")
        for any_hit in (False, True):
            prim, t, u, v, st = dev.intersect(to, td, tmin, tmax, any_hit=any_hit, stats=True)
            torch.cuda.synchronize()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            for _ in range(a.reps):
                dev.intersect(to, td, tmin, tmax, any_hit=any_hit)
            ev1.record()
            torch.cuda.synchronize()
            ms = ev0.elapsed_time(ev1) / a.reps
            hits = int((prim >= 0).sum())
            b_ray = 52 + 8 * st.inner / n + 8 * st.leaves / n + 4 * st.refs / n + 36 * st.prim_tests / n
            print("%-10s %-7s n=%d hits=%d  %.3f ms  %.1f Mrays/s | per ray: inner %.1f leaves %.2f refs %.2f tests %.2f  B_ray %.0f B -> %.1f GB/s" % (
                name, "any" if any_hit else "closest", n, hits, ms, n / ms / 1e3, st.inner / n, st.leaves / n,
                st.refs / n, st.prim_tests / n, b_ray, b_ray * n / ms / 1e6), flush=True)
    dev.close()


if __name__ == "__main__":
    main()
