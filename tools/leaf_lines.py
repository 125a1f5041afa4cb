"""How many 128-byte lines a leaf's primitives span: with 64-byte records in first-reference order and an
index list per leaf in an array of its own (the layout this tool was written to question), against a copy
of every record per reference, contiguous per leaf.  Leaves are weighted by the surface area of their boxes (the chance that a random ray visits them).

usage: python tools/leaf_lines.py [scene] [width height]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import scenes  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "million_bunny"
    sc = getattr(scenes, name)(1920, 1080)
    tree = sc.initKdTree(16)
    d = tree.dump()
    leaf = d["leaf"].astype(bool)
    cnt, off, refs = d["ref_cnt"][leaf], d["ref_off"][leaf], d["refs"]
    ext = d["bmax"][leaf] - d["bmin"][leaf]
    area = 2 * (ext[:, 0] * ext[:, 1] + ext[:, 1] * ext[:, 2] + ext[:, 0] * ext[:, 2])
    n_obj = int(refs.max()) + 1
    print("%s: %d objects, %d nodes, %d leaves (%d empty), %d references (x%.2f)" % (
        name, n_obj, len(leaf), leaf.sum(), (cnt == 0).sum(), len(refs), len(refs) / n_obj))
    # leaves in node (pre-)order; first-reference order of the objects
    order = np.argsort(off, kind="stable")
    cnt, off, area = cnt[order], off[order], area[order]
    seq = np.concatenate([refs[o:o + c] for o, c in zip(off, cnt)]) if len(off) else refs
    _, first = np.unique(seq, return_index=True)
    new_index = np.empty(n_obj, np.int64)
    new_index[seq[np.sort(first)]] = np.arange(len(first))
    hist = np.bincount(np.minimum(cnt, 16), minlength=17)
    w = area / area.sum()
    print("references per leaf: " + " ".join("%d:%.1f%%" % (k, 100 * hist[k] / len(cnt)) for k in range(17)))
    print("  area-weighted mean %.2f (plain mean %.2f)" % ((w * cnt).sum(), cnt.mean()))
    cur_prim = np.zeros(len(cnt)); cur_ref = np.zeros(len(cnt)); dup = np.zeros(len(cnt)); dup48 = np.zeros(len(cnt))
    pos = 0  # position of the leaf's first copy in the per-reference array
    for i, (o, c) in enumerate(zip(off.tolist(), cnt.tolist())):
        if c:
            r = refs[o:o + c]
            cur_prim[i] = len(np.unique(new_index[r] >> 1))
            cur_ref[i] = 0 if c == 1 else (((o + c - 1) * 4) >> 7) - (((o + 1) * 4) >> 7) + 1  # the first reference rides in the node record
            dup[i] = (((pos + c) * 64 - 1) >> 7) - ((pos * 64) >> 7) + 1
            dup48[i] = (((pos + c) * 48 - 1) >> 7) - ((pos * 48) >> 7) + 1
            pos += c
    print("lines per leaf visit, area-weighted, if every primitive of a visited leaf is tested and nothing is cached:")
    print("  64-B records in first-reference order + index list in its own array: records %.2f + list %.2f = %.2f" % (
        (w * cur_prim).sum(), (w * cur_ref).sum(), (w * (cur_prim + cur_ref)).sum()))
    print("  a 64-B copy per reference, contiguous per leaf: %.2f   (%.0f MB instead of %.0f MB)" % (
        (w * dup).sum(), len(refs) * 64 / 1e6, (n_obj * 64 + len(refs) * 4) / 1e6))
    print("  a 48-B copy per reference, contiguous per leaf: %.2f   (%.0f MB)" % ((w * dup48).sum(), len(refs) * 48 / 1e6))


if __name__ == "__main__":
    main()
