"""Differential run of mrt_intersect against the oracle on the seeded random scenes of
tests/fuzz_scenes.py; prints every mismatch.  usage: fuzz_debug.py first_seed n_seeds"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd.host import DeviceScene
from tests import orc, fuzz_scenes
first, count = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 1)
total = 0
for seed in range(first, first + count):
    sc = fuzz_scenes.make(seed)
    tree = sc.initKdTree(1 + seed % 4)
    flat = tree.flatten(seed % 2)
    d_ = flat.desc.contents
    kinds = np.ctypeslib.as_array(d_.obj_kind, shape=(d_.n_objects,))
    dev = DeviceScene(flat)
    o, d, tmin, tmax = fuzz_scenes.rays(flat, seed, n_random=20000)
    o2, d2, tmin2, tmax2 = fuzz_scenes.rays_hard(flat, seed)
    o, d, tmin, tmax = np.concatenate([o, o2]), np.concatenate([d, d2]), np.concatenate([tmin, tmin2]), np.concatenate([tmax, tmax2])
    for any_hit in (False, True):
        want, ws = orc.oracle_intersect(flat.desc, o, d, tmin, tmax, any_hit, stats=True)
        prim, t, u, v, gs = dev.intersect(o, d, tmin, tmax, any_hit=any_hit, stats=True)
        hit = want["prim"] >= 0
        # the kernels without counters are other instantiations of the same loop: same four arrays
        plain = dev.intersect(o, d, tmin, tmax, any_hit=any_hit)
        for x, y in zip(plain, (prim, t, u, v)):
            if not np.array_equal(x.view(np.uint32) if x.dtype == np.float32 else x, y.view(np.uint32) if y.dtype == np.float32 else y):
                total += 1
                print("seed", seed, "any_hit", any_hit, "the kernels with and without counters differ")
                break
        bad = np.nonzero(((prim >= 0) != hit) | (hit & ((t.view(np.uint32) != want["t"].view(np.uint32)) | (prim != want["prim"]))))[0]
        stats_ok = (gs.inner, gs.leaves, gs.refs, gs.prim_tests) == (ws.inner, ws.leaves, ws.refs, ws.prim_tests)
        total += len(bad) + (0 if stats_ok else 1)
        if len(bad) or not stats_ok:
            print("seed", seed, "any_hit", any_hit, "mismatches", len(bad), "of", int(hit.sum()), "stats equal", stats_ok)
        for i in bad[:5]:
            p = prim[i]
            print("  ray", i, "prim gpu/orc", p, want["prim"][i], "kind", kinds[max(p, 0)], "t gpu %r orc %r" % (float(t[i]), float(want["t"][i])), "tmin %g tmax %g" % (tmin[i], tmax[i]), "o", o[i], "d", d[i])
    dev.close()
print("seeds", first, "..", first + count - 1, "total mismatches", total)
