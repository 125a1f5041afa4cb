"""The workloads BASELINE.json names, assembled through the host API exactly as the reference's
example scenes assemble theirs (example_scenes.cpp), with repository-local asset paths.

  cornell      scene-9 family: cornell-box.obj + its MTL lights, 60 deg camera (example_scenes.cpp:155-167)
  bunny_ibl    scene-20 family: stanford-bunny.obj (pre-scaled) diffuse+glossy, Plane floor, HDR env
               (example_scenes.cpp:1371-1459; the reference's 4k map is not distributed, the 1k one is)
  veach_mi     scene 21: four sphere emitters, four rough plates, floor (example_scenes.cpp:1461-1592)
  stress       the bunny mesh 1->4 subdivided `subdivisions` times: the ~1M-triangle kd-tree stress case
  plant        indoor_plant_02.obj + MTL (one material per triangle) with its colour map, HDR env
"""
import os

import numpy as np

from . import abi
from .host import Scene, Transform

ASSETS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets")


def asset(name):
    return os.path.join(ASSETS, name)


def parse_scene(scene_id, width=1200, height=800):
    """parseScene(sceneId) of the reference (example_scenes.cpp:1594-1647) for its Cook-Torrance scenes
    9, 16, 17, 18, 19, 20 and 21, read from assets/scenes/scene<N>.scene; 1200x800 are the reference's WIDTH and
    HEIGHT (example_scenes.h:4-5)."""
    return Scene.parse(scene_id, width, height, ASSETS)


def cornell(width=512, height=512):
    sc = Scene()
    frm, at = (-0.23, 2.585, 5.3), (-0.23, 2.585, -2.8)
    focus = float(np.float32(5.3) - np.float32(-2.8))
    sc.setSimpleCamera(frm, at, (0, 0.1, 0), 60, width, height, 0.001, focus)
    mat = sc.CreateMaterial(abi.MRT_MAT_DIFFUSE, albedo=(0.0, 1.0, 0.95), roughness=0.004, metalness=0.0)
    sc.addObjectsFromFile(asset("cornell-box.obj"), mat, Transform())
    return sc


def bunny_ibl(width=1920, height=1080, glossy=False, obj="stanford-bunny.obj", prescale=20.0, subdivisions=0,
              env="brown_photostudio_06_1k.hdr"):
    """Bunny on a Plane floor under an HDR environment.

    stanford-bunny.obj is in metres (about 0.15 across); at that scale every triangle fails the
    reference's absolute |det| < 1e-4 rejection (triangle.cpp:16), so the vertices are multiplied by
    `prescale` when loading.
    """
    sc = Scene()
    # mesh bounds after x20: x in [-1.9, 1.2], y in [0.66, 3.7], z in [-1.2, 1.2]
    frm, at = (-0.3, 2.6, 6.0), (-0.3, 2.0, 0.0)
    focus = float(np.linalg.norm(np.array(at, np.float32) - np.array(frm, np.float32)).astype(np.float32))
    sc.setSimpleCamera(frm, at, (0, 1, 0), 60, width, height, 0.02, focus)
    if env:
        sc.add_ibl_file(asset(env), also_env=True)
    floor = sc.CreateMaterial(abi.MRT_MAT_DIFFUSE, albedo=(0.6, 0.6, 0.6), metalness=0.0, roughness=0.8)
    t = Transform().scale(8, 8, 1).rotate((-1, 0, 0), 90).translate((0, 0.66, 0))
    sc.add_plane(t, floor)
    if glossy:   # mat5 of scene 20
        body = sc.CreateMaterial(abi.MRT_MAT_SPECULAR, albedo=(1, 1, 1), roughness=0.2, metalness=0.9)
    else:        # mat4 of scene 20
        body = sc.CreateMaterial(abi.MRT_MAT_DIFFUSE, albedo=(0, 1, 0), roughness=0.5, metalness=0.1)
    sc.addObjectsFromFile(asset(obj), body, Transform(), prescale=prescale, subdivisions=subdivisions)
    return sc


def bunny_pair(width=1920, height=1080, env="brown_photostudio_06_1k.hdr", subdivisions=0):
    """BASELINE config 2: stanford-bunny.obj twice -- one diffuse (scene-20 mat4), one glossy
    (scene-20 mat5), each placed by its own Transform as the reference places its bunnies
    (example_scenes.cpp:1439-1457) -- on a Plane floor under the 1k HDR environment."""
    sc = Scene()
    frm, at = (0.0, 2.8, 7.5), (0.0, 2.0, 0.0)
    focus = float(np.linalg.norm(np.array(at, np.float32) - np.array(frm, np.float32)).astype(np.float32))
    sc.setSimpleCamera(frm, at, (0, 1, 0), 60, width, height, 0.02, focus)
    if env:
        sc.add_ibl_file(asset(env), also_env=True)
    floor = sc.CreateMaterial(abi.MRT_MAT_DIFFUSE, albedo=(0.6, 0.6, 0.6), metalness=0.0, roughness=0.8)
    sc.add_plane(Transform().scale(10, 10, 1).rotate((-1, 0, 0), 90).translate((0, 0.66, 0)), floor)
    diffuse = sc.CreateMaterial(abi.MRT_MAT_DIFFUSE, albedo=(0, 1, 0), roughness=0.5, metalness=0.1)
    glossy = sc.CreateMaterial(abi.MRT_MAT_SPECULAR, albedo=(1, 1, 1), roughness=0.2, metalness=0.9)
    sc.addObjectsFromFile(asset("stanford-bunny.obj"), diffuse, Transform().translate((-1.5, 0.0, 0)), prescale=20.0,
                          subdivisions=subdivisions)
    sc.addObjectsFromFile(asset("stanford-bunny.obj"), glossy, Transform().translate((2.1, 0.0, 0)), prescale=20.0,
                          subdivisions=subdivisions)
    return sc


def mixed_kinds(width=1920, height=1080, env="brown_photostudio_06_1k.hdr"):
    """bunny_pair with all three kinds of material the shade kernel distinguishes: a glass bunny
    (TRANSPARENT), a diffuse one (plain Cook-Torrance) and a checker-textured floor."""
    sc = Scene()
    frm, at = (0.0, 2.8, 7.5), (0.0, 2.0, 0.0)
    focus = float(np.linalg.norm(np.array(at, np.float32) - np.array(frm, np.float32)).astype(np.float32))
    sc.setSimpleCamera(frm, at, (0, 1, 0), 60, width, height, 0.02, focus)
    if env:
        sc.add_ibl_file(asset(env), also_env=True)
    floor = sc.CreateMaterial(abi.MRT_MAT_DIFFUSE, albedo=(0.6, 0.6, 0.6), metalness=0.0, roughness=0.8)
    sc.set_checker(floor, (0.8, 0.8, 0.8), (0.15, 0.15, 0.25), Transform().scale(0.05, 0.05, 1))
    sc.add_plane(Transform().scale(10, 10, 1).rotate((-1, 0, 0), 90).translate((0, 0.66, 0)), floor)
    diffuse = sc.CreateMaterial(abi.MRT_MAT_DIFFUSE, albedo=(0, 1, 0), roughness=0.5, metalness=0.1)
    glass = sc.CreateMaterial(abi.MRT_MAT_TRANSPARENT, albedo=(1, 1, 1), ior=1.5, roughness=0.01)
    sc.addObjectsFromFile(asset("stanford-bunny.obj"), diffuse, Transform().translate((-1.5, 0.0, 0)), prescale=20.0)
    sc.addObjectsFromFile(asset("stanford-bunny.obj"), glass, Transform().translate((2.1, 0.0, 0)), prescale=20.0)
    return sc


def million_bunny(width=1920, height=1080, env="brown_photostudio_06_1k.hdr"):
    """BASELINE config 3 size class (the reference's dragon.obj is not available): the stanford
    bunny subdivided twice, 1 111 216 triangles.  Everything is four times the size of bunny_ibl so
    that the subdivided triangles stay clear of the absolute |det| < 1e-4 rejection."""
    sc = Scene()
    frm, at = (-1.2, 10.4, 24.0), (-1.2, 8.0, 0.0)
    focus = float(np.linalg.norm(np.array(at, np.float32) - np.array(frm, np.float32)).astype(np.float32))
    sc.setSimpleCamera(frm, at, (0, 1, 0), 60, width, height, 0.08, focus)
    if env:
        sc.add_ibl_file(asset(env), also_env=True)
    floor = sc.CreateMaterial(abi.MRT_MAT_DIFFUSE, albedo=(0.6, 0.6, 0.6), metalness=0.0, roughness=0.8)
    sc.add_plane(Transform().scale(32, 32, 1).rotate((-1, 0, 0), 90).translate((0, 2.64, 0)), floor)
    body = sc.CreateMaterial(abi.MRT_MAT_SPECULAR, albedo=(1, 1, 1), roughness=0.2, metalness=0.9)
    sc.addObjectsFromFile(asset("stanford-bunny.obj"), body, Transform(), prescale=80.0, subdivisions=2)
    return sc


def read_ppm(path):
    """Binary PPM (P6, 8 bit) as float32 [h][w][3] in 0..1: the values image_texture keeps
    (8-bit channel / 255, textures.hpp:64-76)."""
    with open(path, "rb") as f:
        data = f.read()
    fields, pos = [], 0
    while len(fields) < 4:
        while data[pos:pos + 1].isspace():
            pos += 1
        if data[pos:pos + 1] == b"#":
            pos = data.index(b"\n", pos) + 1
            continue
        end = pos
        while not data[end:end + 1].isspace():
            end += 1
        fields.append(data[pos:end])
        pos = end
    if fields[0] != b"P6" or int(fields[3]) != 255:
        raise ValueError("%s: not an 8-bit binary PPM" % path)
    w, h = int(fields[1]), int(fields[2])
    px = np.frombuffer(data, np.uint8, w * h * 3, pos + 1).reshape(h, w, 3)
    return px.astype(np.float32) / np.float32(255.0)


def plant(width=3840, height=2160, env="brown_photostudio_06_1k.hdr", textured=True, prescale=4.0):
    """BASELINE config 5: indoor_plant_02.obj (22 920 quads = 45 840 triangles) with its MTL file,
    HDR environment.  addObjectsFromFile makes one Material per triangle from the MTL entry of its
    face (example_scenes.cpp:85-92): Kd as albedo, roughness 1, metalness 0.  The reference leaves
    the texture assignment commented out (example_scenes.cpp:111); `textured` attaches the plant's
    colour map (the 1024x1024 version, stored as PPM) to every one of those materials, all sharing
    one copy of the texels.  Vertices are scaled by `prescale` at load: at the file's own scale a
    tenth of the triangles fall under the absolute |det| < 1e-4 rejection (triangle.cpp:16)."""
    sc = Scene()
    frm, at = (-2.0, 22.0, 62.0), (-2.0, 15.0, 0.0)
    focus = float(np.linalg.norm(np.array(at, np.float32) - np.array(frm, np.float32)).astype(np.float32))
    sc.setSimpleCamera(frm, at, (0, 1, 0), 40, width, height, 0.05, focus)
    if env:
        sc.add_ibl_file(asset(env), also_env=True)
    default = sc.CreateMaterial(abi.MRT_MAT_DIFFUSE, albedo=(0.64, 0.64, 0.64), roughness=1.0, metalness=0.0)
    first = sc.n_materials
    sc.addObjectsFromFile(asset("indoor_plant_02.obj"), default, Transform(), prescale=prescale)
    last = sc.n_materials - 1
    if textured and last >= first:
        sc.set_image_texture(first, read_ppm(asset("indoor_plant_2_COL_1k.ppm")))
        if last > first:
            sc.share_texture(first, first + 1, last)
    return sc


def small_bunny(width=256, height=256, env_pixels=None, with_floor=True):
    """bunny.obj (4 968 triangles, about 1.5 units tall): the quick parity scene."""
    sc = Scene()
    frm, at = (0.1, 1.1, 3.4), (0.1, 0.7, 0.0)
    focus = float(np.linalg.norm(np.array(at, np.float32) - np.array(frm, np.float32)).astype(np.float32))
    sc.setSimpleCamera(frm, at, (0, 1, 0), 50, width, height, 0.01, focus)
    if env_pixels is not None:
        sc.add_ibl_pixels(env_pixels, also_env=True)
    if with_floor:
        floor = sc.CreateMaterial(abi.MRT_MAT_DIFFUSE, albedo=(0.6, 0.6, 0.6), metalness=0.0, roughness=0.8)
        t = Transform().scale(5, 5, 1).rotate((-1, 0, 0), 90).translate((0, -0.0148, 0))
        sc.add_plane(t, floor)
    body = sc.CreateMaterial(abi.MRT_MAT_DIFFUSE, albedo=(0, 1, 0), roughness=0.5, metalness=0.1)
    sc.addObjectsFromFile(asset("bunny.obj"), body, Transform(), prescale=1.0)
    return sc


def veach_mi(width=1920, height=1080):
    sc = Scene()
    frm, at = (0, 2, 15), (0, -2, 2.15)
    focus = float(np.linalg.norm(np.array(at, np.float32) - np.array(frm, np.float32)).astype(np.float32))
    sc.setSimpleCamera(frm, at, (0, 1, 0), 28, width, height, 0.001, focus)
    # The first sphere light of scene 21 (radius .5 at (10,10,4)) is created but never added.
    for radius, x, intensity in ((0.1, -1.25, 100.0), (0.03333, -3.75, 901.803), (0.3, 1.25, 11.1111), (0.9, 3.75, 1.23457)):
        m = sc.CreateMaterial(abi.MRT_MAT_DIFFUSE, albedo=(1, 1, 1), emission=(intensity,) * 3)
        t = Transform().scale(radius, radius, radius).translate((x, 0.0, 0))
        sc.add_sphere(t, m)
        sc.add_sphere_light(t, m, (intensity,) * 3)
    for name, rough in (("plate1", 0.005), ("plate2", 0.02), ("plate3", 0.05), ("plate4", 0.1)):
        m = sc.CreateMaterial(abi.MRT_MAT_DIFFUSE, albedo=(0.07, 0.09, 0.13), roughness=rough, metalness=0.0)
        sc.addObjectsFromFile(asset("veach_mi/%s.obj" % name), m, Transform())
    m = sc.CreateMaterial(abi.MRT_MAT_DIFFUSE, albedo=(0.4, 0.4, 0.4), roughness=1.0, metalness=0.0)
    sc.addObjectsFromFile(asset("veach_mi/floor.obj"), m, Transform())
    return sc


def synthetic_env(width=64, height=32, seed=7):
    """A small deterministic HDR environment (smooth sky gradient + one bright lobe)."""
    rng = np.random.default_rng(seed)
    v = (np.arange(height, dtype=np.float32) + 0.5) / height
    u = (np.arange(width, dtype=np.float32) + 0.5) / width
    sky = 0.3 + 0.7 * (1.0 - v)[:, None] * np.ones((1, width), np.float32)
    lobe = 20.0 * np.exp(-(((u[None, :] - 0.3) / 0.05) ** 2 + ((v[:, None] - 0.25) / 0.08) ** 2))
    img = np.stack([sky + lobe, 0.9 * sky + 0.9 * lobe, 1.1 * sky + 0.7 * lobe], axis=-1)
    img += 0.02 * rng.random(img.shape, dtype=np.float32)
    return np.ascontiguousarray(img, dtype=np.float32)
