"""Compares single device shading functions (mrt_debug_eval) with the CPU oracle."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import scenes, abi
from raytracer_amd.host import DeviceScene
from tests import orc
vp = C.c_void_p
lib = orc.oracle()
lib.orc_ibl.argtypes = [C.POINTER(abi.SceneDesc), C.c_int64, vp, vp]
lib.orc_random_dir.argtypes = [C.c_int64, vp, vp, vp]
lib.orc_material_sample.argtypes = [C.POINTER(abi.SceneDesc), C.c_int, C.c_int64] + [vp] * 7
lib.orc_engine_draws.argtypes = [C.c_uint32] * 5 + [C.c_int, vp]
P = lambda a: a.ctypes.data_as(vp)
rng = np.random.default_rng(5)

def unit(n):
    v = rng.normal(size=(n, 3)).astype(np.float32)
    return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)

def report(name, a, b):
    d = np.abs(a - b); rel = d / (np.abs(b) + 1e-6)
    print("%-28s max abs %.3e  max rel %.3e  exact %d/%d" % (name, d.max(), rel.max(), int((a.view(np.uint32) == b.view(np.uint32)).all(axis=-1).sum()) if a.ndim > 1 else int((a.view(np.uint32) == b.view(np.uint32)).sum()), len(a)))

n = 4096
sc = scenes.small_bunny(32, 32, env_pixels=scenes.synthetic_env())
flat = sc.initKdTree().flatten(1); dev = DeviceScene(flat)
# IBL
d = unit(n)
g = dev.debug_eval(abi_ids := 4, d, 7)
o = np.zeros((n, 7), np.float32); lib.orc_ibl(flat.desc, n, P(d), P(o))
report("ibl Le", g[:, 0:3], o[:, 0:3]); report("ibl sample texel", g[:, 3:6], o[:, 3:6]); report("ibl pdf_Li", g[:, 6], o[:, 6])
# random_dir
nr = unit(n); seq = rng.integers(0, 2**63, size=n, dtype=np.uint64)
rows = np.zeros((n, 5), np.float32); rows[:, 0:3] = nr; rows[:, 3] = (seq & 0xffffffff).astype(np.uint32).view(np.float32); rows[:, 4] = (seq >> 32).astype(np.uint32).view(np.float32)
g = dev.debug_eval(5, rows, 3)
o = np.zeros((n, 3), np.float32); lib.orc_random_dir(n, P(nr), P(seq), P(o))
report("random_dir", g, o)
# sampler
rows = np.zeros((n, 5), np.uint32); rows[:, 0] = 7; rows[:, 1] = rng.integers(0, 10000, n); rows[:, 2] = 4; rows[:, 3] = 4; rows[:, 4] = rng.integers(0, 16, n)
g = dev.debug_eval(6, rows.view(np.float32), 33)
o = np.zeros((n, 33), np.float32)
for i in range(n):
    cam = np.zeros(5, np.float32); a = np.zeros(8, np.float32); b = np.zeros(16, np.float32); e = np.zeros(4, np.float32)
    lib.orc_sample_streams(int(rows[i, 0]), int(rows[i, 1]), 4, 4, int(rows[i, 4]), P(cam), P(a), P(b))
    lib.orc_engine_draws(int(rows[i, 0]), int(rows[i, 1]), 4, 4, int(rows[i, 4]), 4, P(e))
    o[i] = np.concatenate([cam, a, b, e])
report("sampler streams", g, o)
# material sample / eval (material 1 = bunny body, 0 = floor)
for mat in (0, 1):
    rd = unit(n); nrm = unit(n)
    flip = (rd * nrm).sum(1) > 0; nrm[flip] = -nrm[flip]
    out = np.ones(n, np.uint8)
    rows = np.zeros((n, 10), np.float32); rows[:, 0] = mat; rows[:, 1:4] = rd; rows[:, 4:7] = nrm; rows[:, 7] = 1
    rows[:, 8] = (seq & 0xffffffff).astype(np.uint32).view(np.float32); rows[:, 9] = (seq >> 32).astype(np.uint32).view(np.float32)
    g = dev.debug_eval(1, rows, 7)
    ob, op, ow = np.zeros((n, 3), np.float32), np.zeros(n, np.float32), np.zeros((n, 3), np.float32)
    lib.orc_material_sample(flat.desc, mat, n, P(rd), P(nrm), P(out), P(seq), P(ob), P(op), P(ow))
    report("material_sample bsdf m%d" % mat, g[:, 0:3], ob); report("material_sample pdf", g[:, 3], op); report("material_sample wi", g[:, 4:7], ow)
dev.close()
# veach: sphere lights
sc = scenes.veach_mi(32, 32)
flat = sc.initKdTree().flatten(1); dev = DeviceScene(flat)
for light in range(4):
    pos = (rng.random((n, 3), dtype=np.float32) * 8 - 4).astype(np.float32); pos[:, 1] = -np.abs(pos[:, 1])
    nrm = unit(n); u = rng.random((n, 2), dtype=np.float32); outside = np.ones(n, np.uint8)
    rows = np.zeros((n, 10), np.float32); rows[:, 0] = light; rows[:, 1:4] = pos; rows[:, 4:7] = nrm; rows[:, 7] = 1; rows[:, 8:10] = u
    g = dev.debug_eval(2, rows, 7)
    op, on, opdf = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32), np.zeros(n, np.float32)
    lib.orc_light_shape_sample(flat.desc, light, n, P(pos), P(nrm), P(outside), P(u), P(op), P(on), P(opdf))
    report("light %d sample pos" % light, g[:, 0:3], op); report("light sample nrm", g[:, 3:6], on); report("light sample pdf", g[:, 6], opdf)
    wi = unit(n)
    rows = np.zeros((n, 10), np.float32); rows[:, 0] = light; rows[:, 1:4] = pos; rows[:, 4:7] = nrm; rows[:, 7:10] = wi
    g = dev.debug_eval(3, rows, 1)
    opdf = np.zeros(n, np.float32); lib.orc_light_shape_pdf(flat.desc, light, n, P(pos), P(nrm), P(wi), P(opdf))
    report("light pdf", g[:, 0], opdf)
dev.close()
