"""terra_amd_set_job_order: LDS-resident launches hand out the pixel blocks that some camera ray hits first and the empty ones last (so that the launch ends on short
jobs). Only the order in which jobs are taken changes, so the framebuffer must be the same bit for bit -- with and without the order, against the oracle, for whole
frames, rectangles, shards and host-framebuffer tile calls; what a job computes is the reference's terra_render / terra_trace (src/Terra.c:551-572, 1039-1097)."""
import ctypes as C

import numpy as np
import pytest

from terra_amd import api, runtime, scenes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L(amd_lib):
    lib = runtime.load()
    assert lib.device_count() > 0, "gpu tests need a visible MI355X: " + runtime.last_error()
    return lib


def dev(L, d, order, split=1, passes=1, shard=None, rect=None, counters=False):
    import torch
    scene = scenes.build_scene(L, d, counters=counters)
    assert L.set_job_order(scene, int(order)) == 0 and L.get_job_order(scene) == int(order)      # 2: launches of any size (1, the default, leaves those below 256 pixel blocks alone)
    assert L.set_sample_split(scene, split) == 0
    fb = runtime.DeviceFramebuffer(d.width, d.height); cam = scenes.camera_of(d)
    rc = torch.zeros(d.width * d.height, dtype=torch.int32, device="cuda") if counters else None
    for _ in range(passes):
        if shard:
            tile, world = shard
            for rank in range(world):
                runtime.render_device_sharded(L, cam, scene, fb, tile, rank, world)
        else:
            runtime.render_device(L, cam, scene, fb, rect, rc)
    torch.cuda.synchronize()
    assert runtime.last_error() == ""
    res = fb.results_host()
    out = dict(pixels=fb.pixels_host().copy(), acc=res["acc"].copy(), samples=res["samples"].copy(), calls=rc.cpu().numpy().copy() if counters else None)
    L.scene_destroy(scene)
    return out


def same_fb(a, b):
    return (np.array_equal(a["acc"].view(np.uint32), b["acc"].view(np.uint32)) and np.array_equal(a["samples"], b["samples"])
            and np.array_equal(a["pixels"].view(np.uint32), b["pixels"].view(np.uint32)) and (a["calls"] is None or np.array_equal(a["calls"], b["calls"])))


# a camera far enough back that a good share of the frame's blocks see nothing: both classes, and blocks on the silhouette
WIDE = dict(camera_position=(0.0, 1.0, -7.0))


@pytest.mark.parametrize("integ", [api.kTerraIntegratorSimple, api.kTerraIntegratorDirect, api.kTerraIntegratorDirectMis, api.kTerraIntegratorDebugNormals])
@pytest.mark.parametrize("counters", [False, True])
def test_order_does_not_change_the_frame(L, integ, counters):
    for mk, split, passes in ((lambda: scenes.cornell_box(328, 200, 8, integrator=integ, **WIDE), 1, 2),
                              (lambda: scenes.cornell_phong(264, 152, 16, integrator=integ, tonemap=api.kTerraTonemappingOperatorReinhard), 4, 1)):      # ragged sizes
        on = dev(L, mk(), 2, split=split, passes=passes, counters=counters)
        off = dev(L, mk(), 0, split=split, passes=passes, counters=counters)
        assert same_fb(on, off), (integ, counters, split)
        assert (on["samples"] == mk().spp * passes).all()


def test_ordered_frame_equals_the_oracle(H, L, orc_lib):
    mk = lambda: scenes.cornell_box(160, 96, 4, integrator=api.kTerraIntegratorDirect, **WIDE)
    H.set_oracle_math(1)
    try:
        want = H.Unit("orc").render_pixels(mk(), passes=2, want_calls=False)
    finally:
        H.set_oracle_math(0)
    got = dev(L, mk(), 2, passes=2)
    assert np.array_equal(got["acc"].view(np.uint32), np.ascontiguousarray(want["acc"]).view(np.uint32))
    assert np.array_equal(got["pixels"].view(np.uint32), np.ascontiguousarray(want["pixels"]).view(np.uint32))
    assert (np.abs(got["acc"]).sum(axis=-1) == 0).mean() > 0.2          # (the scene really leaves part of the frame empty)


def test_rectangles_shards_and_small_launches(L):
    mk = lambda: scenes.cornell_box(456, 264, 8, integrator=api.kTerraIntegratorSimple, **WIDE)
    for kw in (dict(rect=(40, 24, 300, 200)), dict(rect=(0, 0, 48, 48)), dict(shard=(64, 3)), dict(shard=(64, 8), split=4), dict(split=8)):
        on = dev(L, mk(), 2, **kw)
        off = dev(L, mk(), 0, **kw)
        assert same_fb(on, off), kw


def test_host_framebuffer_tiles_from_threads(L):
    # terra_render() on a host framebuffer, tile calls from several threads (the reference client's pattern, satellite/src/Renderer.cpp:70-98): the order lives in the thread's scratch
    import threading
    d = scenes.cornell_box(512, 320, 16, integrator=api.kTerraIntegratorDirect, **WIDE)
    frames = []
    for order in (2, 0):
        scene = scenes.build_scene(L, d, counters=False); cam = scenes.camera_of(d)
        L.set_job_order(scene, order); L.set_sample_split(scene, 0)
        fb = api.Framebuffer(L, d.width, d.height)
        jobs = [(x, y, min(128, d.width - x), min(128, d.height - y)) for y in range(0, d.height, 128) for x in range(0, d.width, 128)]

        def worker(k):
            for j in jobs[k::4]:
                L.render(C.byref(cam), scene, C.byref(fb.fb), *j)
        ths = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
        [t.start() for t in ths]; [t.join() for t in ths]
        assert runtime.last_error() == ""
        frames.append((fb.pixels.copy(), fb.results["acc"].copy(), fb.results["samples"].copy()))
        fb.destroy(); L.scene_destroy(scene)
    (p2, a2, s2), (p1, a1, s1) = frames
    assert np.array_equal(p2.view(np.uint32), p1.view(np.uint32)) and np.array_equal(a2.view(np.uint32), a1.view(np.uint32)) and np.array_equal(s2, s1)


def test_default_leaves_small_launches_alone_and_rejects_other_values(L):
    mk = lambda: scenes.cornell_box(200, 120, 8, **WIDE)          # 13 x 8 = 104 pixel blocks
    assert same_fb(dev(L, mk(), 1), dev(L, mk(), 0))
    scene = scenes.build_scene(L, mk(), counters=False)
    assert L.get_job_order(scene) == 1 and L.set_job_order(scene, 3) != 0 and L.get_job_order(scene) == 1
    runtime.load().clear_error()
    L.scene_destroy(scene)


def test_full_size_headline_frame(L):
    # BASELINE.json configs[1]: 1920 x 1080, 512 spp, split 32, as bench.py times it
    mk = lambda: scenes.cornell_box(1920, 1080, 512, bounces=8, integrator=api.kTerraIntegratorSimple)
    on = dev(L, mk(), 2, split=32)
    off = dev(L, mk(), 0, split=32)
    assert same_fb(on, off)
    assert (on["samples"] == 512).all()
