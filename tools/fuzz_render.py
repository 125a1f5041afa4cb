"""mrt_render against the oracle's path tracer (same counter-based samples) on the random shaded
scenes of tests/fuzz_scenes.py.  usage: fuzz_render.py first_seed n_seeds [min_roughness [rich]]"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd.host import DeviceScene, render_params
from tests import orc, fuzz_scenes
first, count = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 1)
min_roughness = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
rich = len(sys.argv) > 4 and sys.argv[4] == "rich"
W, H, SX, SY = 64, 48, 2, 2
worst = 0.0
for seed in range(first, first + count):
    sc = fuzz_scenes.make_shaded(seed, min_roughness, rich)
    flat = sc.initKdTree(2).flatten(1)
    dev = DeviceScene(flat)
    p = render_params(W, H, SX, SY, max_depth=6, seed=seed + 1)
    gpu = np.zeros((H, W, 3), np.float32)
    rs = dev.render(p, gpu)
    cpu, crs, _, _ = orc.oracle_render(flat.desc, p, 0, threads=16)
    dev.close()
    g, c = gpu / (SX * SY), cpu / (SX * SY)
    # a pixel that is not finite in one must be not finite in the other (NaN where NaN, the same infinity)
    gbad, cbad = ~np.isfinite(g).all(axis=2), ~np.isfinite(c).all(axis=2)
    same_bad = int((gbad != cbad).sum())
    ok = ~(gbad | cbad)
    gg, cc = g[ok], c[ok]
    rel = float(np.mean((gg - cc) ** 2 / (cc ** 2 + 1e-2)))
    frac = float(((np.abs(gg - cc) / (np.abs(cc) + 1e-3)).max(axis=1) <= 1e-4).mean())
    flag = "" if (same_bad == 0 and rel <= 1e-5 and frac >= 0.995) else "   <-- outside the tolerance"
    worst = max(worst, rel)
    print("seed %3d lights/objects %d/%d not finite gpu/cpu/differ %d/%d/%d relMSE %.3e pixels within 1e-4: %.4f rays gpu/cpu %d/%d%s" % (
        seed, sc.n_lights, sc.n_objects, gbad.sum(), cbad.sum(), same_bad, rel, frac, rs.rays_closest + rs.rays_any, crs.rays_closest + crs.rays_any, flag), flush=True)
print("worst relMSE", worst)
