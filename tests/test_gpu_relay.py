"""The relay of long rays (device_trace.hpp: Relay) leaves every result as it was.

A ray still in flight some hundred trips after its launch's list ran out is handed over at the next
leaf boundary and the rest of its interval is traced by a follow-up launch as MRT_RELAY_K shorter rays
side by side.  With MRT_RELAY_AFTER=0 a wave hands its rays over at the first leaf boundary after its
list ran out -- in the small batches of a test that is most rays that live beyond their first leaf --
so these tests push the bulk of their rays through the hand-over, the segment launch and the
completion kernel, and compare, bit for bit, with the CPU oracle and with the same library run with
MRT_RELAY=0.
"""
import numpy as np
import pytest

from raytracer_amd import abi, scenes
from raytracer_amd.host import DeviceScene, render_params
from tests import orc
from tests.test_gpu_intersect import _bits, _compare, _many_transform_scene, _rays

pytestmark = pytest.mark.gpu


def _scene(name):
    if name == "cornell":  # emissive triangles: the renderer's shadow rays are not relayed there
        return scenes.cornell(96, 96)
    if name == "bunny_plane":
        return scenes.small_bunny(96, 96, env_pixels=scenes.synthetic_env())
    if name == "veach":  # spheres, sphere emitters
        return scenes.veach_mi(96, 54)
    if name == "transforms":  # many transforms, two planes, spheres
        return _many_transform_scene()
    raise KeyError(name)


@pytest.mark.parametrize("after", [0, 7])
@pytest.mark.parametrize("name", ["cornell", "bunny_plane", "veach", "transforms"])
def test_relayed_rays_against_the_oracle_and_against_no_relay(built, monkeypatch, name, after):
    sc = _scene(name)
    flat = sc.initKdTree().flatten(1)
    o, d, tmin, tmax = _rays(flat, 96, 96, 30000, 17 + after)
    monkeypatch.setenv("MRT_RELAY", "0")
    plain = DeviceScene(flat)
    monkeypatch.delenv("MRT_RELAY")
    monkeypatch.setenv("MRT_RELAY_AFTER", str(after))
    dev = DeviceScene(flat)
    for any_hit in (False, True):
        before = dev.relayed()
        assert _compare(flat, dev, o, d, tmin, tmax, any_hit) > 1000
        a = dev.intersect(o, d, tmin, tmax, any_hit=any_hit)
        b = plain.intersect(o, d, tmin, tmax, any_hit=any_hit)
        for x, y in zip(a, b):  # prim, t, u, v of every ray, misses included
            assert np.array_equal(_bits(x) if x.dtype == np.float32 else x, _bits(y) if y.dtype == np.float32 else y)
        moved = dev.relayed() - before
        if any_hit:
            assert moved == 0  # mrt_intersect reports WHICH occluder an any-hit ray met: those are traced in one piece
        elif name != "transforms":  # (whose 190 objects make a tree of a handful of leaves: hardly a deferred node to resume at)
            assert moved > 2000, "only %d rays went through the relay" % moved
    assert plain.relayed() == 0
    plain.close()
    dev.close()


@pytest.mark.parametrize("seed", list(range(8)))
def test_relayed_rays_in_random_scenes_with_ties(built, monkeypatch, seed):
    """The tie-producing random scenes and their hard rays (from vertices, along edges, -0.0 components,
    empty and inverted intervals), with the hand-over forced."""
    from tests import fuzz_scenes
    monkeypatch.setenv("MRT_RELAY_AFTER", "0")
    sc = fuzz_scenes.make(100 + seed)
    flat = sc.initKdTree().flatten(seed & 1)
    dev = DeviceScene(flat)
    o, d, tmin, tmax = fuzz_scenes.rays(flat, seed)
    o2, d2, tmin2, tmax2 = fuzz_scenes.rays_hard(flat, seed)
    o, d, tmin, tmax = np.concatenate([o, o2]), np.concatenate([d, d2]), np.concatenate([tmin, tmin2]), np.concatenate([tmax, tmax2])
    for any_hit in (False, True):
        assert _compare(flat, dev, o, d, tmin, tmax, any_hit) > 500
    want = orc.oracle_intersect(flat.desc, o, d, tmin, tmax, False)
    prim, t, u, v = dev.intersect(o, d, tmin, tmax)
    kinds = np.ctypeslib.as_array(flat.desc.contents.obj_kind, shape=(sc.n_objects,))
    tri = (prim >= 0) & (kinds[np.maximum(prim, 0)] == 3)
    assert np.array_equal(u[tri], want["u"][tri]) and np.array_equal(v[tri], want["v"][tri])
    assert dev.relayed() > 100
    dev.close()


@pytest.mark.parametrize("name", ["cornell", "bunny_plane", "veach"])
def test_frames_with_and_without_the_relay_are_the_same_bits(built, monkeypatch, name):
    """mrt_render over several passes, overlapped and serial: every traversal launch of every bounce
    hands rays over; the frame equals the one rendered with MRT_RELAY=0 bit for bit, and the ray counts
    agree."""
    sc = _scene(name)
    w, h = (96, 54) if name == "veach" else (96, 96)
    flat = sc.initKdTree().flatten(1)
    frames = []
    for relay in (False, True, True):
        if relay:
            monkeypatch.delenv("MRT_RELAY", raising=False)
            monkeypatch.setenv("MRT_RELAY_AFTER", "0")
        else:
            monkeypatch.setenv("MRT_RELAY", "0")
        dev = DeviceScene(flat)
        img = np.zeros((h, w, 3), np.float32)
        flags = abi.MRT_RENDER_SERIAL if len(frames) == 2 else 0
        rs = dev.render(render_params(w, h, 4, 2, max_depth=5, seed=3, paths_per_pass=w * h * 3, flags=flags), img)
        frames.append((img, rs.rays_closest, rs.rays_any, dev.relayed()))
        dev.close()
    assert frames[0][3] == 0 and frames[1][3] > 1000 and frames[2][3] > 1000  # (how many: depends on which wave drew which chunk)
    for f in frames[1:]:
        assert np.array_equal(f[0], frames[0][0]) and f[1:3] == frames[0][1:3]
    assert frames[0][0].max() > 0


@pytest.mark.parametrize("name", ["cornell", "bunny_plane", "veach"])
def test_camera_rays_as_packets_give_the_same_frame(built, monkeypatch, name):
    """The camera rays of a pixel walking the tree together (trace_packets, MRT_CAMERA_PACKETS=1), through
    the streaming traversal (=0), or whichever the library's timing picks (unset): the same frame bit for
    bit.  With MRT_INTERSECT_PACKETS=1 mrt_intersect's closest-hit batches go through the packet traversal:
    hit ids and distances equal the oracle's on camera rays, random rays and short rays, and all four
    output arrays equal the streaming traversal's."""
    sc = _scene(name)
    w, h = (96, 54) if name == "veach" else (96, 96)
    flat = sc.initKdTree().flatten(1)
    frames = []
    for mode in ("0", "1", None):
        if mode is None:
            monkeypatch.delenv("MRT_CAMERA_PACKETS", raising=False)
        else:
            monkeypatch.setenv("MRT_CAMERA_PACKETS", mode)
        dev = DeviceScene(flat)
        img = np.zeros((h, w, 3), np.float32)
        rs = dev.render(render_params(w, h, 8, 8, max_depth=4, seed=6, paths_per_pass=w * h * 40), img)
        choice = dev.camera_choice()
        assert choice[0] == int(mode) if mode is not None else (choice[0] in (0, 1) and choice[1] > 0 and choice[2] > 0)
        frames.append((img, rs.rays_closest, rs.rays_any))
        dev.close()
    for f in frames[1:]:
        assert np.array_equal(f[0], frames[0][0]) and f[1:] == frames[0][1:]
    plain = DeviceScene(flat)
    monkeypatch.setenv("MRT_INTERSECT_PACKETS", "1")
    dev = DeviceScene(flat)
    o, d, tmin, tmax = _rays(flat, 96, 96, 30000, 23)
    assert _compare(flat, dev, o, d, tmin, tmax, False) > 1000
    for x, y in zip(dev.intersect(o, d, tmin, tmax), plain.intersect(o, d, tmin, tmax)):  # prim, t, u, v of every ray, misses included
        assert np.array_equal(_bits(x) if x.dtype == np.float32 else x, _bits(y) if y.dtype == np.float32 else y)
    dev.close()
    plain.close()


@pytest.mark.parametrize("name", ["cornell", "bunny_plane", "veach", "transforms"])
def test_the_mailbox_changes_nothing(built, monkeypatch, name):
    """A ray remembers the last triangles its line was found to miss and passes them over, unfetched, in the
    leaves that reference them again (trace_stream: mailbox).  With MRT_MAILBOX=0 every primitive of every
    visited leaf is fetched and tested: hits, distances, barycentrics, the traversal counters (a triangle
    passed over counts as tested, as the reference tests it) and frames are the same bits.  cornell's walls
    are two triangles each that dozens of leaves reference and rays hit after leaves in which the same
    triangle's distance lay outside the leaf's interval: those rejections must not be remembered."""
    sc = _scene(name)
    w, h = (96, 54) if name == "veach" else (96, 96)
    flat = sc.initKdTree().flatten(1)
    o, d, tmin, tmax = _rays(flat, 96, 96, 30000, 31)
    out = []
    for mode in (None, "0"):
        if mode is None:
            monkeypatch.delenv("MRT_MAILBOX", raising=False)
        else:
            monkeypatch.setenv("MRT_MAILBOX", mode)
        dev = DeviceScene(flat)
        res = []
        for any_hit in (False, True):
            assert _compare(flat, dev, o, d, tmin, tmax, any_hit) > 1000
            r = dev.intersect(o, d, tmin, tmax, any_hit=any_hit, stats=True)
            res.append(r[:4] + ((r[4].inner, r[4].leaves, r[4].refs, r[4].prim_tests),))
        img = np.zeros((h, w, 3), np.float32)
        rs = dev.render(render_params(w, h, 4, 4, max_depth=5, seed=9, paths_per_pass=w * h * 5), img)
        out.append((res, img, rs.rays_closest, rs.rays_any))
        dev.close()
    for a, b in zip(out[0][0], out[1][0]):
        assert a[4] == b[4]
        for x, y in zip(a[:4], b[:4]):
            assert np.array_equal(_bits(x) if x.dtype == np.float32 else x, _bits(y) if y.dtype == np.float32 else y)
    assert np.array_equal(out[0][1], out[1][1]) and out[0][2:] == out[1][2:]
    assert name == "transforms" or out[0][1].max() > 0  # (that scene has no light)
