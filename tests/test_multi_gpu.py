"""Several GPUs from one process behind the C-ABI (include/terra_amd.h: terra_amd_set_devices, terra_amd_render_multi, terra_amd_shard_owner,
terra_amd_multi_info): the layout of the reference's client -- one process, tiles dealt to workers (satellite/src/Renderer.cpp:316-350).

CPU: the tile -> device map for 2, 4 and 8 devices, argument checking. GPU (one device on the test box): the multi-device code path with a device set
of ONE -- replica bookkeeping, communicator, pack, the RCCL gather (issued, not short-circuited), unpack, one copy to the host -- against terra_render()
and the oracle, bit for bit; and a REHEARSAL of 2, 3 and 5 replicas on the one device (terra_amd_debug_replicas_share_device): every replica its own copy of the
scene with rebased pointers, its own stream and share of the tiles, the gather a stand-in (RCCL admits one rank per device). The RCCL transport between DISTINCT
devices has never run (no multi-GPU box is available to the build): DESIGN.md "Multi-GPU" says so.
"""
import ctypes as C

import numpy as np
import pytest

from terra_amd import api, runtime, scenes


@pytest.fixture(scope="module")
def L(amd_lib):
    return runtime.load(need_torch=False)


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_tile_to_device_map(L, world):
    """every tile of a frame has exactly one owner, the owners' tile counts are what terra_amd_shard_tile_count says and differ by at most one, and the map is the
    kernels' shard rule t % world (DevRenderParams::rank / world)"""
    for (w, h, tile) in ((1920, 1080, 64), (3840, 2160, 64), (100, 70, 32), (64, 64, 64)):
        tx, ty = -(-w // tile), -(-h // tile)
        owners = np.array([L.shard_owner(t, world) for t in range(tx * ty)])
        assert owners.min() >= 0 and owners.max() < world and np.array_equal(owners, np.arange(tx * ty) % world)
        counts = [int((owners == r).sum()) for r in range(world)]
        assert counts == [L.fn("terra_amd_shard_tile_count", C.c_int, [C.c_size_t] * 3 + [C.c_int] * 2)(w, h, tile, r, world) for r in range(world)]
        assert max(counts) - min(counts) <= 1 and sum(counts) == tx * ty
    assert L.shard_owner(5, 0) < 0
    L.clear_error()


def test_device_set_arguments(L):
    n = L.device_count()
    one = (C.c_int * 1)(0)
    got = (C.c_int * 4)()
    assert L.get_devices(got, 4) == 1                                   # default: the one device of terra_amd_set_device
    if n == 0:
        assert L.set_devices(one, 1) < 0 and "not available" in runtime.last_error()
    else:
        assert L.set_devices(one, 1) == 0 and L.get_devices(got, 4) == 1 and got[0] == 0
        two = (C.c_int * 2)(0, 0)
        assert L.set_devices(two, 2) < 0 and "twice" in runtime.last_error()
        bad = (C.c_int * 1)(n)
        assert L.set_devices(bad, 1) < 0
    assert L.set_devices(None, 3) < 0                                   # a count without a list
    assert L.set_devices(None, 0) == 0 and L.get_devices(got, 4) == 1   # back to the single device
    L.clear_error()


@pytest.mark.gpu
def test_multi_device_path_with_one_device_equals_terra_render_and_the_oracle(H, orc_lib, devmath_mode):
    import torch  # noqa: F401
    Lg = runtime.load()
    one = (C.c_int * 1)(0)
    for d, tile, rect in ((scenes.cornell_box(200, 136, 4, integrator=api.kTerraIntegratorDirect), 64, None),
                          (scenes.cornell_spheres(160, 96, 2), 32, (16, 8, 112, 80))):
        want = H.Unit("orc").render_pixels(d, passes=2, rect=rect)
        Lg.clear_error()
        runtime.check(Lg.set_devices(one, 1))
        try:
            scene = scenes.build_scene(Lg, d)
            assert runtime.last_error() == ""
            fb = api.Framebuffer(Lg, d.width, d.height); cam = scenes.camera_of(d)
            x, y, w, h = rect or (0, 0, d.width, d.height)
            before = runtime.MultiInfo(); runtime.check(Lg.multi_info(scene, C.byref(before)))
            for _ in range(2):                                          # progressive: the second call accumulates on the first one's sums
                runtime.check(Lg.render_multi(C.byref(cam), scene, C.byref(fb.fb), x, y, w, h, tile))
            info = runtime.MultiInfo(); runtime.check(Lg.multi_info(scene, C.byref(info)))
            assert info.devices == 1 and info.device[0] == 0 and info.replicas == 1
            assert info.gathers == 2 and info.process_collectives >= before.process_collectives + 2        # the collective was ISSUED both times
            assert info.rccl_version > 0 and info.communicator_ranks == 1 and b"rccl" in info.rccl_library
            tiles = -(-w // tile) * -(-h // tile)
            assert info.last_gather_bytes == tiles * tile * tile * 28
            assert np.array_equal(H.bits(fb.results["acc"]), H.bits(want["acc"])) and np.array_equal(fb.results["samples"], want["samples"])
            nan = np.isnan(want["pixels"])
            assert np.array_equal(H.bits(fb.pixels)[~nan], H.bits(want["pixels"])[~nan])
            # ... and the drop-in entry on the same scene (a set of one device takes the ordinary path) gives the same frame
            fb2 = api.Framebuffer(Lg, d.width, d.height)
            for _ in range(2):
                Lg.render(C.byref(cam), scene, C.byref(fb2.fb), x, y, w, h)
            assert runtime.last_error() == ""
            assert np.array_equal(H.bits(fb2.results["acc"]), H.bits(fb.results["acc"]))
            fb.destroy(); fb2.destroy(); Lg.scene_destroy(scene)
        finally:
            Lg.set_devices(None, 0)


@pytest.mark.gpu
def test_rehearsal_of_several_replicas_on_one_device(H, orc_lib, devmath_mode):
    """2, 3 and 5 replicas of a scene on device 0: what several devices run, except the transport. Scenes chosen for what a replica has to carry: the fast tree and
    its triangle soup, a texture (descriptors rebased), the environment's sampling tables (a second allocation), the light tables of Direct + MIS."""
    import torch  # noqa: F401
    from test_oracle_vs_reference import soup_scene
    Lg = runtime.load()
    rs = np.random.RandomState(5)
    big = soup_scene(H, 1500, 91, integrator=2); big.width, big.height, big.spp = 136, 100, 2
    env = scenes.cornell_box(132, 72, 2, integrator=1)
    env.environment_texture = scenes.TextureDesc(rs.uniform(0, 3, size=(6, 11, 3)).astype(np.float32), address_mode=0); env.environment_lighting = True; env.environment_sampling = True
    tex = scenes.cornell_textured(96, 64, 2) if hasattr(scenes, "cornell_textured") else scenes.cornell_phong(96, 64, 2, integrator=1)
    runtime.check(Lg.debug_replicas_share_device(1))
    try:
        for d in (big, env, tex):
            want = H.Unit("orc").render_pixels(d, passes=2, want_calls=False)
            for world in (2, 3, 5):
                devs = (C.c_int * world)(*([0] * world))
                Lg.clear_error()
                runtime.check(Lg.set_devices(devs, world), "terra_amd_set_devices")
                scene = scenes.build_scene(Lg, d)
                assert runtime.last_error() == "", runtime.last_error()
                fb = api.Framebuffer(Lg, d.width, d.height); cam = scenes.camera_of(d)
                before = runtime.MultiInfo(); runtime.check(Lg.multi_info(scene, C.byref(before)))
                assert before.devices == world and before.replicas == world
                for _ in range(2):
                    runtime.check(Lg.render_multi(C.byref(cam), scene, C.byref(fb.fb), 0, 0, d.width, d.height, 32), runtime.last_error())
                info = runtime.MultiInfo(); runtime.check(Lg.multi_info(scene, C.byref(info)))
                assert info.gathers == 2 and info.rehearsed_gathers >= before.rehearsed_gathers + 2 and info.process_collectives == before.process_collectives      # (the stand-in, and it says so)
                assert np.array_equal(H.bits(fb.results["acc"]), H.bits(want["acc"])) and np.array_equal(fb.results["samples"], want["samples"]), (d.name, world)
                nan = np.isnan(want["pixels"])
                assert np.array_equal(H.bits(fb.pixels)[~nan], H.bits(want["pixels"])[~nan]), (d.name, world)
                fb.destroy(); Lg.scene_destroy(scene)
        # the drop-in entry on a scene with two replicas: a call that covers enough of the frame is sharded over them (include/terra_amd.h terra_amd_set_devices),
        # a tile-sized call goes to one replica -- the calling thread's -- whole
        d = scenes.cornell_box(512, 288, 1, integrator=1)
        want = H.Unit("orc").render_pixels(d, want_calls=False)
        runtime.check(Lg.set_devices((C.c_int * 2)(0, 0), 2))
        scene = scenes.build_scene(Lg, d); cam = scenes.camera_of(d)
        fb = api.Framebuffer(Lg, d.width, d.height)
        g0 = runtime.MultiInfo(); runtime.check(Lg.multi_info(scene, C.byref(g0)))
        Lg.render(C.byref(cam), scene, C.byref(fb.fb), 0, 0, d.width, d.height)
        g1 = runtime.MultiInfo(); runtime.check(Lg.multi_info(scene, C.byref(g1)))
        assert runtime.last_error() == "" and g1.gathers == g0.gathers + 1
        assert np.array_equal(H.bits(fb.results["acc"]), H.bits(want["acc"]))
        fb2 = api.Framebuffer(Lg, d.width, d.height)
        import threading
        def tiles(rows):
            for ty in rows:
                for tx in range(0, d.width, 128):
                    Lg.render(C.byref(cam), scene, C.byref(fb2.fb), tx, ty, min(128, d.width - tx), min(96, d.height - ty))
        th = [threading.Thread(target=tiles, args=(range(k * 96, d.height, 192),)) for k in range(2)]          # two client threads: each lands on its own replica
        [t.start() for t in th]; [t.join() for t in th]
        g2 = runtime.MultiInfo(); runtime.check(Lg.multi_info(scene, C.byref(g2)))
        assert runtime.last_error() == "" and g2.gathers == g1.gathers              # tile-sized calls are not sharded
        assert np.array_equal(H.bits(fb2.results["acc"]), H.bits(want["acc"]))
        fb.destroy(); fb2.destroy(); Lg.scene_destroy(scene)
    finally:
        Lg.set_devices(None, 0); Lg.debug_replicas_share_device(0)
    two = (C.c_int * 2)(0, 0)
    assert Lg.set_devices(two, 2) < 0 and "twice" in runtime.last_error()          # the hook is off again
    Lg.clear_error()
