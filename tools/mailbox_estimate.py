"""How often a ray meets the same primitive again in a later leaf: walks the kd-tree on the CPU for a
sample of rays leaving the camera rays' hit points (no primitive tests: the walk lists the references of
every leaf the ray's interval reaches, up to the distance of the ray's own hit), and counts what a small
per-ray memory of recently tested primitives would have caught.

usage: python tools/mailbox_estimate.py [scene] [rays]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import scenes  # noqa: E402
from tests import orc  # noqa: E402


def walk(d, o, dr, t0, t1):
    """leaf reference lists in visiting order (the reference's interval logic, without NaN care)"""
    out = []
    stack = [(0, t0, t1)]
    leaf, axis, split, left, right, off, cnt, refs = (d[k] for k in ("leaf", "axis", "split", "left", "right", "ref_off", "ref_cnt", "refs"))
    while stack:
        n, a, b = stack.pop()
        while not leaf[n]:
            ax = axis[n]
            diff = split[n] - o[ax]
            t = diff / dr[ax] if dr[ax] != 0 else np.inf * np.sign(diff) if diff != 0 else np.nan
            near, far = (left[n], right[n]) if diff > 0 else (right[n], left[n])
            if t > b or t < 0:
                n = near
            elif t < a:
                n = far
            else:
                stack.append((far, t, b))
                n, b = near, t
        out.append(refs[off[n]:off[n] + cnt[n]])
    return out


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "million_bunny"
    n_rays = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
    W, H = 1920, 1080
    sc = getattr(scenes, name)(W, H)
    tree = sc.initKdTree(16)
    flat = tree.flatten(1)
    d = tree.dump()
    rng = np.random.default_rng(3)
    o, dr = orc.camera_rays(flat.desc, W, H, seed=1)
    pick = rng.choice(len(o), 4 * n_rays, replace=False)
    o, dr = o[pick], dr[pick]
    far = np.full(len(o), 1e5, np.float32)
    h = orc.oracle_intersect(flat.desc, o, dr, np.zeros(len(o), np.float32), far, False)
    hit = h["prim"] >= 0
    p = (o + h["t"][:, None] * dr)[hit][:n_rays]
    w = rng.normal(size=p.shape).astype(np.float32)
    w /= np.linalg.norm(w, axis=1, keepdims=True)
    flip = (w * dr[hit][:n_rays]).sum(1) > 0
    w[flip] = -w[flip]
    o1 = (p + np.float32(1e-3) * w).astype(np.float32)
    h1 = orc.oracle_intersect(flat.desc, o1, w, np.zeros(len(o1), np.float32), far[:len(o1)], False)
    bmin, bmax = d["bmin"][0], d["bmax"][0]
    # the library's object order: first reference, leaves in node order
    lf = d["leaf"].astype(bool)
    order = np.argsort(d["ref_off"][lf], kind="stable")
    seq_all = np.concatenate([d["refs"][o:o + c] for o, c in zip(d["ref_off"][lf][order], d["ref_cnt"][lf][order])])
    _, first = np.unique(seq_all, return_index=True)
    new_index = np.zeros(int(seq_all.max()) + 1, np.int64)
    new_index[seq_all[np.sort(first)]] = np.arange(len(first))
    lines2 = lines3 = lines4 = 0
    tests = distinct = leaves = 0
    caught = {1: 0, 2: 0, 4: 0, 8: 0}
    caught_prev_leaf = 0
    pol = {"lru": 0, "again": 0}
    for i in range(len(o1)):
        with np.errstate(divide="ignore", invalid="ignore"):
            ta, tb = (bmin - o1[i]) / w[i], (bmax - o1[i]) / w[i]
        t0, t1 = max(0.0, np.minimum(ta, tb).max()), min(1e5, np.maximum(ta, tb).min())
        if t0 > t1:
            continue
        lists = walk(d, o1[i].astype(np.float64), w[i].astype(np.float64), t0, t1)
        if h1["prim"][i] >= 0:  # the search ends in the first leaf that holds the hit primitive with the hit inside: take the first that lists it
            for k, l in enumerate(lists):
                if h1["prim"][i] in l:
                    lists = lists[:k + 1]
                    break
        seq = np.concatenate(lists) if lists else np.zeros(0, np.int32)
        leaves += len(lists)
        tests += len(seq)
        distinct += len(np.unique(seq))
        lines2 += len(np.unique(new_index[seq] >> 1))
        lines3 += len(np.unique(new_index[seq] // 3))
        lines4 += len(np.unique(new_index[seq] >> 2))
        for nmb in caught:
            box = []
            for x in seq:
                if x in box:
                    caught[nmb] += 1
                else:
                    box.append(x)
                    if len(box) > nmb:
                        box.pop(0)
        # policies of an 8-entry memory: true LRU, and first-in first-out where a caught primitive is entered again
        lru, dup = [], []
        for x in seq:
            if x in lru:
                pol["lru"] += 1
                lru.remove(x)
            lru.append(x)
            if len(lru) > 8:
                lru.pop(0)
            if x in dup:
                pol["again"] += 1
            dup.append(x)
            if len(dup) > 8:
                dup.pop(0)
        prev = set()
        for l in lists:
            caught_prev_leaf += sum(1 for x in l if x in prev)
            prev = set(l.tolist())
    n = len(o1)
    print("%s, %d rays from the camera rays' hit points: leaves %.2f  tests %.2f  distinct primitives %.2f per ray" % (
        name, n, leaves / n, tests / n, distinct / n))
    print("  distinct lines of primitive records per ray: %.2f at two records a line (in use), %.2f at three, %.2f at four" % (lines2 / n, lines3 / n, lines4 / n))
    for nmb, c in caught.items():
        print("  a memory of the last %d distinct primitives catches %.1f %% of the tests" % (nmb, 100 * c / tests))
    print("  8 entries, least recently used replaced: %.1f %%; first-in first-out with caught primitives entered again: %.1f %%" % (
        100 * pol["lru"] / tests, 100 * pol["again"] / tests))
    print("  'was in the leaf before' catches %.1f %%" % (100 * caught_prev_leaf / tests))


if __name__ == "__main__":
    main()
