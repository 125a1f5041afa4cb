"""Writes assets/scenes/scene19.scene: scene 19 of the reference (example_scenes.cpp:1276-1369) places
two 7x7 grids of spheres in loops whose float counters accumulate double increments
(`roughness += 0.14` on a float), so the values are produced here with the same arithmetic and
written with nine significant digits, which round-trip a float exactly."""
import math
import os

import numpy as np

f32 = np.float32


def g(x):
    return "%.9g" % float(x)


def add(x, inc):   # float x; x += <double literal>
    return f32(float(x) + inc)


out = ["# Scene 19 of the reference (example_scenes.cpp:1276-1369): two 7x7 grids of spheres, roughness growing",
       "# along one axis and metalness along the other, under an HDR environment.  Written by",
       "# tools/make_scene19.py (the reference computes the positions and parameters in float loops).  The",
       "# reference names brown_photostudio_06_4k.hdr, which is not distributed; this is the 1k version.",
       "camera from -3.5 -2 7.2 at 3.0 -2.4 2.6 up 0 1 0 fov 80 lens 0.04 focus distance",
       "env brown_photostudio_06_1k.hdr"]
n = 0
for grid in range(2):
    metalness, y = f32(0.01), f32(-6.4)
    for i in range(7):
        x, roughness = (f32(-2) if grid == 0 else f32(8.6)), f32(0.01)
        for j in range(7):
            name = "s%d" % n
            n += 1
            if grid == 0:
                out.append("material %s roughness %s metalness %s albedo 0.9 0 0" % (name, g(roughness), g(metalness)))
                out.append("transform scale 0.5 0.5 0.5 translate %s %s 0" % (g(x), g(y)))
            else:
                out.append("material %s type specular roughness %s metalness %s albedo 0.8 0.1 0" % (name, g(roughness), g(metalness)))
                out.append("transform scale 0.5 0.5 0.5 translate 6 %s %s" % (g(y), g(x)))
            out.append("sphere %s" % name)
            roughness = add(roughness, 0.14)
            x = add(x, 1.2) if grid == 0 else add(x, -1.2)
        metalness = add(metalness, 0.14)
        y = add(y, 1.2)
# matE: created, its sphere is commented out.  0.09 * 4.f * M_PI / (4. * M_PI * lightSize * lightSize * M_PI), lightSize = 0.01f
ls = float(f32(0.01))
e = f32(0.09 * 4.0 * math.pi / (4. * math.pi * ls * ls * math.pi))
out.append("material matE emission %s %s %s albedo 1 1 1" % (g(e), g(e), g(e)))
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets", "scenes", "scene19.scene")
open(path, "w").write("\n".join(out) + "\n")
print(path, n, "spheres")
