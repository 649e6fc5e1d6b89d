"""terra_amd_set_sample_split: S lanes per pixel. One call of n spp with split S must leave bit for bit the
framebuffer (sums, sample counts, tonemapped pixels, rand-call totals) of S successive calls of n/S spp --
on the device and, through the oracle / the goldens' accumulate semantics, in the reference."""
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


def dev(L, d, split=1, passes=1, shard=None, rect=None):
    import torch
    scene = scenes.build_scene(L, d)
    assert L.set_sample_split(scene, split) == 0 and L.get_sample_split(scene) == split
    fb = runtime.DeviceFramebuffer(d.width, d.height); cam = scenes.camera_of(d)
    rc = torch.zeros(d.width * d.height, dtype=torch.int32, device="cuda")
    tot = np.zeros((d.height, d.width), np.uint64)
    for _ in range(passes):
        if shard:
            tile, world = shard
            for rank in range(world):
                runtime.render_device_sharded(L, cam, scene, fb, tile, rank, world)
        else:
            rc.zero_()
            runtime.render_device(L, cam, scene, fb, rect, rc)
            torch.cuda.synchronize()
            tot += rc.cpu().numpy().astype(np.uint32).reshape(d.height, d.width)
    torch.cuda.synchronize()
    res = fb.results_host()
    out = dict(pixels=fb.pixels_host().copy(), acc=res["acc"].copy(), samples=res["samples"].copy(), rand_calls=tot)
    L.scene_destroy(scene)
    return out


def same_fb(a, b):
    return (np.array_equal(a["acc"].view(np.uint32), b["acc"].view(np.uint32)) and np.array_equal(a["samples"], b["samples"])
            and np.array_equal(a["pixels"].view(np.uint32), b["pixels"].view(np.uint32)))


def with_spp(d, spp):
    d.spp = spp
    return d


@pytest.mark.parametrize("split", [2, 4, 8, 16])
def test_split_equals_successive_calls(H, L, split):
    mk = lambda spp: scenes.cornell_phong(104, 72, spp, integrator=api.kTerraIntegratorDirect, tonemap=api.kTerraTonemappingOperatorReinhard)   # ragged: 104 = 6.5 blocks
    one = dev(L, mk(32), split=split)
    many = dev(L, mk(32 // split), split=1, passes=split)
    assert same_fb(one, many)
    assert np.array_equal(one["rand_calls"], many["rand_calls"])
    assert (one["samples"] == 32).all()


def test_split_against_the_oracle_and_accumulation(H, L, orc_lib):
    d = scenes.cornell_box(64, 48, 16)
    H.set_oracle_math(1)
    try:
        want = H.Unit("orc").render_pixels(with_spp(scenes.cornell_box(64, 48, 4), 4), passes=4 + 2)
    finally:
        H.set_oracle_math(0)
    # one split call of 16 (= 4 x 4) on top of 2 plain calls of 4... order matters: do the plain ones first on both sides
    import torch
    scene = scenes.build_scene(L, with_spp(scenes.cornell_box(64, 48, 4), 4))
    fb = runtime.DeviceFramebuffer(64, 48); cam = scenes.camera_of(d)
    runtime.render_device(L, cam, scene, fb); runtime.render_device(L, cam, scene, fb)
    o = L.scene_get_options(scene).contents; o.samples_per_pixel = 16; L.scene_commit(scene)
    L.set_sample_split(scene, 4)
    runtime.render_device(L, cam, scene, fb); torch.cuda.synchronize()
    res = fb.results_host()
    assert np.array_equal(res["acc"].view(np.uint32), want["acc"].view(np.uint32)) and (res["samples"] == 24).all()
    assert np.array_equal(fb.pixels_host().view(np.uint32), want["pixels"].view(np.uint32))
    L.scene_destroy(scene)


def test_automatic_split_is_a_function_of_the_call(H, L):
    # 64x48 frame = 12 blocks: automatic picks 16 lanes per pixel at 256 spp (16 samples each), 2 at 32 spp, none at 16 spp
    for spp, s in ((256, 16), (32, 2), (16, 1)):
        mk = lambda n: scenes.cornell_box(64, 48, n)
        assert same_fb(dev(L, mk(spp), split=0), dev(L, mk(spp // s), passes=s)), spp


def test_split_falls_back_when_spp_is_not_a_multiple(H, L):
    # 12 spp with split 8 -> 4 chunks of 3; 7 spp -> no split at all
    assert same_fb(dev(L, scenes.cornell_box(48, 48, 12), split=8), dev(L, scenes.cornell_box(48, 48, 3), passes=4))
    assert same_fb(dev(L, scenes.cornell_box(48, 48, 7), split=8), dev(L, scenes.cornell_box(48, 48, 7)))


def test_split_with_shards_rectangles_and_host_entry(H, L):
    mk = lambda spp: scenes.cornell_box(160, 112, spp, integrator=api.kTerraIntegratorDirectMis)
    want = dev(L, mk(2), passes=4)
    assert same_fb(dev(L, mk(8), split=4, shard=(32, 3)), want)
    rect = (24, 16, 100, 71)
    a, b = dev(L, mk(8), split=4, rect=rect), dev(L, mk(2), passes=4, rect=rect)
    assert same_fb(a, b) and not a["samples"][:16].any()
    # terra_render() on a host framebuffer honours the split too
    d = mk(8)
    scene = scenes.build_scene(L, d); L.set_sample_split(scene, 4)
    fb = api.Framebuffer(L, d.width, d.height); cam = scenes.camera_of(d)
    L.clear_error()
    L.render(C.byref(cam), scene, C.byref(fb.fb), 0, 0, d.width, d.height)
    assert runtime.last_error() == ""
    assert np.array_equal(fb.results["acc"].view(np.uint32), want["acc"].view(np.uint32)) and np.array_equal(fb.pixels.view(np.uint32), want["pixels"].view(np.uint32))
    assert L.set_sample_split(scene, 3) < 0 and "sample split" in runtime.last_error()
    L.clear_error()
    fb.destroy(); L.scene_destroy(scene)


def test_threaded_tile_calls_with_automatic_split(H, L):
    """the reference client's pattern: worker threads call terra_render() on disjoint tiles of one framebuffer; with the
    automatic split every call picks its own lanes-per-pixel, and the frame must not depend on threads or call order"""
    import threading
    d = scenes.cornell_box(192, 128, 64, integrator=api.kTerraIntegratorDirect)
    tiles = [(x, y, min(64, d.width - x), min(48, d.height - y)) for y in range(0, d.height, 48) for x in range(0, d.width, 64)]
    frames = []
    for threaded in (False, True):
        scene = scenes.build_scene(L, d); assert L.set_sample_split(scene, 0) == 0
        fb = api.Framebuffer(L, d.width, d.height); cam = scenes.camera_of(d)
        L.clear_error()
        if threaded:
            def work(k):
                for t in tiles[k::4][::-1]:
                    L.render(C.byref(cam), scene, C.byref(fb.fb), *t)
            ths = [threading.Thread(target=work, args=(k,)) for k in range(4)]
            [t.start() for t in ths]; [t.join() for t in ths]
        else:
            for t in tiles:
                L.render(C.byref(cam), scene, C.byref(fb.fb), *t)
        assert runtime.last_error() == ""
        frames.append((fb.results["acc"].copy(), fb.pixels.copy(), fb.results["samples"].copy()))
        fb.destroy(); L.scene_destroy(scene)
    assert np.array_equal(frames[0][0].view(np.uint32), frames[1][0].view(np.uint32)) and np.array_equal(frames[0][1].view(np.uint32), frames[1][1].view(np.uint32))
    assert (frames[0][2] == 64).all()
    # a worker stages its largest TILE, not the frame, and a worker's error reaches the polling thread through the process-wide channel
    seen = {}
    def probe():
        scene = scenes.build_scene(L, d); fb = api.Framebuffer(L, d.width, d.height); cam = scenes.camera_of(d)
        L.render(C.byref(cam), scene, C.byref(fb.fb), 64, 48, 64, 48)
        seen["bytes"] = L.thread_staging_bytes()
        L.render(C.byref(cam), scene, C.byref(fb.fb), 100, 100, 500, 500)          # outside the frame
        fb.destroy(); L.scene_destroy(scene)
    L.clear_first_error(); L.clear_error()
    th = threading.Thread(target=probe); th.start(); th.join()
    assert seen["bytes"] >= 64 * 48 * 28 and seen["bytes"] < d.width * d.height * 28       # (a new thread may inherit the slot of a worker that has ended: never smaller than its own largest tile)
    assert runtime.last_error() == "" and "bad tile rectangle" in runtime.first_error()[1]
    L.clear_first_error()
    # a 64x48 tile is 12 blocks: the automatic rule gives 4 lanes per pixel at 64 spp (16 samples each) = 4 calls of 16
    want = dev(L, scenes.cornell_box(192, 128, 16, integrator=api.kTerraIntegratorDirect), passes=4)
    assert np.array_equal(frames[0][0].view(np.uint32), want["acc"].view(np.uint32))


def test_short_lived_worker_threads_hand_their_slots_on(H, L):
    """a client that starts fresh worker threads for every frame: a thread that ends hands its stream and buffers to the next new thread
    (scene_host.cpp SlotPool); three generations of workers accumulate three passes, identical to three device-resident passes"""
    import threading
    d = scenes.cornell_box(160, 96, 8)
    tiles = [(x, y, min(48, d.width - x), min(32, d.height - y)) for y in range(0, d.height, 32) for x in range(0, d.width, 48)]
    scene = scenes.build_scene(L, d); assert L.set_sample_split(scene, 0) == 0
    fb = api.Framebuffer(L, d.width, d.height); cam = scenes.camera_of(d)
    L.clear_error(); L.clear_first_error()
    sizes = []
    for generation in range(3):
        def work(k):
            for t in tiles[k::5]:
                L.render(C.byref(cam), scene, C.byref(fb.fb), *t)
            sizes.append(L.thread_staging_bytes())
        ths = [threading.Thread(target=work, args=(k,)) for k in range(5)]
        [t.start() for t in ths]; [t.join() for t in ths]
    assert runtime.last_error() == "" and runtime.first_error()[0] == 0
    assert min(sizes) >= 48 * 32 * 28
    got = (fb.results["acc"].copy(), fb.results["samples"].copy())
    fb.destroy(); L.scene_destroy(scene)
    assert (got[1] == 24).all()
    # reference: the same 3 x tiles calls from this one thread
    scene = scenes.build_scene(L, d); assert L.set_sample_split(scene, 0) == 0
    fb = api.Framebuffer(L, d.width, d.height)
    for generation in range(3):
        for t in tiles:
            L.render(C.byref(cam), scene, C.byref(fb.fb), *t)
    assert np.array_equal(fb.results["acc"].view(np.uint32), got[0].view(np.uint32))
    fb.destroy(); L.scene_destroy(scene)
