"""Times mrt_intersect on the ray classes of the benchmark frame (device-resident buffers).

Scene: the bench.py scene (two stanford bunnies, plane floor).  Ray classes:
  primary   camera rays, closest hit
  bounce    rays leaving the primary hit points in random directions of the outgoing hemisphere
            (closest hit) - the shape of the extend rays of bounce >= 1 and of the MIS rays
  shadow    the same rays as any-hit queries with tmax 1e4 - the shape of the environment shadow rays

usage: python tools/trace_bench.py [--reps N] [--layout L] [--small]
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
    ap.add_argument("--layout", type=int, default=1)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--mult", type=int, default=8, help="secondary rays per primary hit")
    ap.add_argument("--subdiv", type=int, default=0, help="midpoint subdivisions of the meshes (x4 triangles each)")
    ap.add_argument("--octant", type=int, default=0, help="group secondary rays by direction octant within runs of this many rays")
    ap.add_argument("--tile", action="store_true", help="order the primary rays in 8x8 pixel tiles")
    ap.add_argument("--second", action="store_true", help="also time second-generation rays, unsorted and sorted by origin")
    ap.add_argument("--small", action="store_true", help="69k-triangle single bunny instead of the bench scene")
    a = ap.parse_args()
    W, H = 1920, 1080
    t0 = time.time()
    sc = scenes.bunny_ibl(W, H, env=None, subdivisions=a.subdiv) if a.small else scenes.bunny_pair(W, H, env=None, subdivisions=a.subdiv)
    tree = sc.initKdTree()
    print("objects %d, tree (nodes, refs, depth) %s, build %.1fs" % (sc.n_objects, tree.counts(), time.time() - t0), flush=True)
    flat = tree.flatten(a.layout)
    dev = DeviceScene(flat)

    from tests import orc
    o, d = orc.camera_rays(flat.desc, W, H, seed=1)
    if a.tile:
        # 8x8 pixel tiles instead of scanlines: 64 consecutive rays cover a square, not a line
        idx = np.arange(W * H).reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(-1)
        o, d = o[idx], d[idx]
    n = len(o)
    to = torch.from_numpy(np.ascontiguousarray(o)).cuda()
    td = torch.from_numpy(np.ascontiguousarray(d)).cuda()
    prim, t, u, v = dev.intersect(to, td, torch.zeros(n, device="cuda"), torch.full((n,), 1e5, device="cuda"))[:4]
    hit = prim >= 0
    g = torch.Generator(device="cuda")
    g.manual_seed(7)
    sos, sds, ws = [], [], []
    for _ in range(a.mult):  # one "sample" after another, pixels in scanline order within each
        w = torch.randn((n, 3), device="cuda", generator=g)
        w = w / w.norm(dim=1, keepdim=True)
        flip = (w * td).sum(1) > 0
        w[flip] = -w[flip]
        ws.append(w)
        sos.append((to + t[:, None] * td + 1e-3 * w)[hit])
        sds.append(w[hit])
    if a.tile:
        # the renderer's order: (tile, sample, pixel in tile)
        def tile_major(parts):
            x = torch.stack(parts).reshape(a.mult, n // 64, 64, -1).permute(1, 0, 2, 3).reshape(a.mult * n, -1)
            return x
        keep = tile_major([hit[:, None].float()] * a.mult)[:, 0] > 0
        allo = tile_major([to + t[:, None] * td + 1e-3 * w_ for w_ in ws])
        alld = tile_major(ws)
        so, sd = allo[keep].contiguous(), alld[keep].contiguous()
    else:
        so = torch.cat(sos).contiguous()
        sd = torch.cat(sds).contiguous()
    if a.octant:
        # within every run of a.octant consecutive rays, group the rays by the signs of their direction
        m = len(so)
        octant = ((sd[:, 0] < 0).long() | ((sd[:, 1] < 0).long() << 1) | ((sd[:, 2] < 0).long() << 2))
        key = (torch.arange(m, device="cuda") // a.octant) * 8 + octant
        order = torch.sort(key, stable=True).indices
        so, sd = so[order].contiguous(), sd[order].contiguous()
    sets = (("primary", to, td, 1e5, False), ("bounce", so, sd, 1e5, False), ("shadow", so, sd, 1e4, True))
    total_ms = 0.0
    for name, oo, dd, far, any_hit in sets:
        m = len(oo)
        tmin = torch.zeros(m, device="cuda")
        tmax = torch.full((m,), far, device="cuda")
        p, _, _, _, st = dev.intersect(oo, dd, tmin, tmax, any_hit=any_hit, stats=True)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(a.reps):
            dev.intersect(oo, dd, tmin, tmax, any_hit=any_hit)
        ev1.record()
        torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / a.reps
        total_ms += ms
        b_ray = 52 + (8 * st.inner + 8 * st.leaves + 4 * st.refs + 36 * st.prim_tests) / m
        digest = int((p.to(torch.int64) * torch.arange(1, m + 1, device="cuda") % 1000003).sum().item())
        print("%-8s n=%d hits=%d  %.3f ms  %.1f Mrays/s | per ray: inner %.1f leaves %.2f refs %.2f tests %.2f  B_ray %.0f B -> %.1f GB/s  digest %d" % (
            name, m, int((p >= 0).sum()), ms, m / ms / 1e3, st.inner / m, st.leaves / m, st.refs / m,
            st.prim_tests / m, b_ray, b_ray * m / ms / 1e6, digest), flush=True)
    print("total %.3f ms" % total_ms)
    if a.second:
        # second-generation rays (leaving the hit points of the bounce rays): in the order of their
        # parents, and sorted by the Morton code of their origin
        m = len(so)
        p2, t2 = dev.intersect(so, sd, torch.zeros(m, device="cuda"), torch.full((m,), 1e5, device="cuda"))[:2]
        h2 = p2 >= 0
        w = torch.randn((m, 3), device="cuda", generator=g)
        w = w / w.norm(dim=1, keepdim=True)
        flip = (w * sd).sum(1) > 0
        w[flip] = -w[flip]
        o2 = (so + t2[:, None] * sd + 1e-3 * w)[h2].contiguous()
        d2 = w[h2].contiguous()
        lo, hi = o2.min(0).values, o2.max(0).values
        q = ((o2 - lo) / (hi - lo + 1e-9) * 1023).long().clamp(0, 1023)
        code = torch.zeros(len(o2), dtype=torch.long, device="cuda")
        for b in range(10):
            for k in range(3):
                code |= ((q[:, k] >> b) & 1) << (3 * b + k)
        order = torch.sort(code).indices
        for name, oo, dd in (("second", o2, d2), ("second, origin-sorted", o2[order].contiguous(), d2[order].contiguous())):
            mm = len(oo)
            tmin = torch.zeros(mm, device="cuda")
            tmax = torch.full((mm,), 1e4, device="cuda")
            st = dev.intersect(oo, dd, tmin, tmax, stats=True)[4]
            torch.cuda.synchronize()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            for _ in range(a.reps):
                dev.intersect(oo, dd, tmin, tmax)
            ev1.record()
            torch.cuda.synchronize()
            ms = ev0.elapsed_time(ev1) / a.reps
            print("%-22s n=%d  %.3f ms  %.1f Mrays/s | per ray: inner %.1f leaves %.2f tests %.2f" % (
                name, mm, ms, mm / ms / 1e3, st.inner / mm, st.leaves / mm, st.prim_tests / mm), flush=True)
        # the same rays eight times over: a list long enough that the kernel's tail does not count
        o8, d8 = o2.repeat(8, 1).contiguous(), d2.repeat(8, 1).contiguous()
        mm = len(o8)
        tmin, tmax = torch.zeros(mm, device="cuda"), torch.full((mm,), 1e4, device="cuda")
        dev.intersect(o8, d8, tmin, tmax)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        dev.intersect(o8, d8, tmin, tmax)
        ev1.record()
        torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1)
        print("%-22s n=%d  %.3f ms  %.1f Mrays/s" % ("second x8", mm, ms, mm / ms / 1e3), flush=True)
    # the same bounce batch handed over as host buffers (PCIe both ways inside the call)
    ho, hd = so.cpu().numpy(), sd.cpu().numpy()
    m = len(ho)
    htmin, htmax = np.zeros(m, np.float32), np.full(m, 1e5, np.float32)
    dev.intersect(ho, hd, htmin, htmax)
    t0 = time.perf_counter()
    dev.intersect(ho, hd, htmin, htmax)
    dt = time.perf_counter() - t0
    print("bounce, host buffers: %.1f ms  %.1f Mrays/s (copies included)" % (dt * 1e3, m / dt / 1e6))
    dev.close()


if __name__ == "__main__":
    main()
