"""Seeded random Wavefront OBJ + MTL files for differential tests of the host library's loader against
the reference's tiny_obj_loader.h (v1.0.6) and the face walk of example_scenes.cpp:32-151.

The files stay inside what the reference defines: every index is valid; when the file has normals
(texture coordinates) every corner names one (the reference indexes attrib.normals with -1 otherwise);
when an MTL library is loaded a known `usemtl` precedes the first face (materials[-1] otherwise).
Inside those limits they vary what real exporters vary: number spellings, white space, line ends,
comments, unknown statements, groups and objects, polygons of 3 to 7 corners, relative (negative)
indices, extra components on v / vt lines, statement order, MTL statements the reference ignores."""
import os

import numpy as np


def _num(rng, lo=-4.0, hi=4.0):
    x = float(rng.uniform(lo, hi))
    how = int(rng.integers(0, 12))
    if rng.random() < 0.02:  # words tinyobj's grammar refuses (they read as 0) or cuts short
        return ["nan", "inf", "-", "+", "1e", "2.5e+", "abc", "-.75", "0x10", "1,5", "3.25f", "7.5.1", "1e5e2"][int(rng.integers(0, 13))]
    if how == 0:
        return "%d" % int(round(x))
    if how == 1:
        return "%.1f" % x
    if how == 2:
        return "%.17g" % x
    if how == 3:
        return "%e" % x
    if how == 4:
        return ("%E" % (x * 1e-3)).replace("E-0", "E-").replace("E+0", "E+")
    if how == 5:
        s = "%.4f" % abs(x)
        return ("-" if x < 0 else "+") + s
    if how == 6:
        s = "%.3f" % (abs(x) % 1.0)
        return ("-" if x < 0 else "") + s[1:]  # ".125"
    if how == 7:
        return "%d." % int(x)
    if how == 8:
        return "-0" if rng.random() < 0.5 else "0"
    if how == 9:
        return "000%.5f" % abs(x)
    if how == 10:
        return "%.9fe%+d" % (x / 100.0, 2)
    return "%.6f" % x


def _ws(rng):
    return [" ", "  ", "\t", " \t "][int(rng.integers(0, 4))]


def make_obj(seed, directory, name="fuzz"):
    """Writes <directory>/<name>.obj (and .mtl for most seeds); returns the OBJ path."""
    rng = np.random.default_rng(seed + 31000)
    eol = ["\n", "\n", "\n", "\r\n", "\r\n", "\r"][int(rng.integers(0, 6))]
    has_vn = rng.random() < 0.5
    has_vt = rng.random() < 0.5
    has_mtl = rng.random() < 0.7
    lines = []

    def junk():
        r = rng.random()
        if r < 0.08:
            lines.append("# a comment " + _num(rng))
        elif r < 0.12:
            lines.append("")
        elif r < 0.15:
            lines.append("s %s" % ("off" if rng.random() < 0.5 else "1"))
        elif r < 0.17:
            lines.append("   # indented comment")

    mtl_names = []
    if has_mtl:
        mtl_lines = ["# material library"]
        if rng.random() < 0.15:  # statements before the first newmtl go to a material that is dropped
            mtl_lines += ["Kd 0.9 0.8 0.7", "Ke 3 3 3"]
        for m in range(int(rng.integers(1, 6))):
            nm = "mat_%d%s" % (m, "" if rng.random() < 0.7 else ".%03d" % m)
            if mtl_names and rng.random() < 0.12:
                nm = mtl_names[0]  # a second material of the same name: stored, never found
            mtl_names.append(nm)
            mtl_lines.append("newmtl" + _ws(rng) + nm)
            stmts = []
            if rng.random() < 0.85:
                stmts.append("Kd %s %s %s" % tuple(_num(rng, 0, 1) for _ in range(3)))
            if rng.random() < 0.5:
                stmts.append("Ks 1 1 1" if rng.random() < 0.4 else "Ks %s %s %s" % tuple(_num(rng, 0, 1) for _ in range(3)))
            if rng.random() < 0.4:
                stmts.append("Ke %s %s %s" % tuple(_num(rng, 0, 20) for _ in range(3)) if rng.random() < 0.7 else "Ke 0 0 0")
            if rng.random() < 0.5:
                stmts.append("Ni %s" % _num(rng, 1, 2))
            for extra in ("Ka 0.1 0.1 0.1", "Ns 96.078431", "d 0.5", "Tr 0.2", "illum 2", "map_Kd texture.png", "Tf 1 1 1", "# note"):
                if rng.random() < 0.3:
                    stmts.append(extra)
            rng.shuffle(stmts)
            for s in stmts:
                mtl_lines.append(("\t" if rng.random() < 0.3 else "") + s.replace(" ", _ws(rng), 1) + (" " if rng.random() < 0.2 else ""))
            mtl_lines.append("")
        with open(os.path.join(directory, name + ".mtl"), "w", newline="") as f:
            f.write(eol.join(mtl_lines) + eol)
        r = rng.random()
        if r < 0.08:
            lines.append("mtllib missing_file.mtl " + name + ".mtl")  # tried in turn
        elif r < 0.12:
            lines.append("mtllib  " + name + ".mtl")  # two blanks: an empty first name
        elif r < 0.16:
            lines.append("mtllib no_such_library.mtl")  # no material at all
        else:
            lines.append("mtllib" + (" " if rng.random() < 0.7 else "\t") + name + ".mtl")

    nv = nn = nt = 0

    def emit_vertices(count):
        nonlocal nv, nn, nt
        for _ in range(count):
            comps = [_num(rng) for _ in range(3)] + ([_num(rng, 0.5, 1.0)] if rng.random() < 0.1 else [])
            lines.append(("  " if rng.random() < 0.1 else "") + "v" + _ws(rng) + _ws(rng).join(comps) + (" " if rng.random() < 0.1 else ""))
            nv += 1
            junk()
        if has_vn:
            for _ in range(max(1, count // 2)):
                lines.append("vn" + _ws(rng) + _ws(rng).join(_num(rng, -1, 1) for _ in range(3)))
                nn += 1
        if has_vt:
            for _ in range(max(1, count // 2)):
                k = 2 if rng.random() < 0.8 else (3 if rng.random() < 0.7 else 1)
                lines.append("vt" + _ws(rng) + _ws(rng).join(_num(rng, -1, 2) for _ in range(k)))
                nt += 1

    def index(n):
        i = int(rng.integers(1, n + 1))
        if rng.random() < 0.01:
            return "0"  # not an OBJ index; tinyobj reads it as the first element
        if rng.random() < 0.03:
            return "+%d" % i
        return str(i - n - 1) if rng.random() < 0.25 else str(i)

    def emit_faces(count):
        for _ in range(count):
            k = int(rng.choice([3, 3, 3, 4, 4, 5, 6, 7]))
            corners = []
            for _c in range(k):
                c = index(nv)
                if has_vt and has_vn:
                    c += "/" + index(nt) + "/" + index(nn)
                elif has_vt:
                    c += "/" + index(nt)
                elif has_vn:
                    c += "//" + index(nn)
                corners.append(c)
            lines.append("f" + _ws(rng) + _ws(rng).join(corners) + ("  " if rng.random() < 0.15 else ""))
            junk()

    used = False
    for block in range(int(rng.integers(1, 5))):
        r = rng.random()
        if r < 0.35:
            lines.append("g" + _ws(rng) + "group%d" % block + (" second_name" if rng.random() < 0.2 else ""))
        elif r < 0.55:
            lines.append("o" + _ws(rng) + "object%d" % block)
        emit_vertices(int(rng.integers(3, 12)))
        if has_mtl and (not used or rng.random() < 0.6):
            lines.append("usemtl" + _ws(rng) + mtl_names[int(rng.integers(0, len(mtl_names)))])
            used = True
        emit_faces(int(rng.integers(1, 8)))
        if has_mtl and rng.random() < 0.4:
            lines.append("usemtl " + mtl_names[int(rng.integers(0, len(mtl_names)))])
            emit_faces(int(rng.integers(1, 4)))
        if rng.random() < 0.15:
            lines.append("g empty_group")
        if has_mtl and rng.random() < 0.2:
            # usemtl straight before g / o: tinyobj 1.0.6 drops the faces the shape holds so far
            lines.append("usemtl " + mtl_names[int(rng.integers(0, len(mtl_names)))])
    path = os.path.join(directory, name + ".obj")
    with open(path, "w", newline="") as f:
        f.write(eol.join(lines) + (eol if rng.random() < 0.8 else ""))
    return path
