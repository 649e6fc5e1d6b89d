"""Environment term. The lat-long lookup is the reference's terra_texture_sample_latlong
(src/Terra.c:468-477) and is pinned to it; ADDING the environment to the image is an extension behind
terra_amd_set_environment_lighting (the reference has that line commented out, src/Terra.c:1056), so renders
with it on are pinned device <-> oracle only, and with it off they stay the reference's (the goldens)."""
import ctypes as C

import numpy as np
import pytest

from terra_amd import api, scenes


def env_textures(H):
    r = H.rng(21)
    return (scenes.TextureDesc(r.uniform(0, 3, size=(8, 16, 3)).astype(np.float32)),
            scenes.TextureDesc(r.randint(0, 256, size=(5, 7, 3)).astype(np.uint8), address_mode=2))


def latlong_all(lib, td, dirs):
    data = np.ascontiguousarray(td.data)
    h, w, c = data.shape
    t = api.TerraTexture()
    (lib.texture_init if data.dtype == np.uint8 else lib.texture_init_hdr)(C.byref(t), w, h, c, data.ctypes.data)
    t.address_mode = td.address_mode; t.filter = td.filter
    out = np.array([lib.texture_sample_latlong(C.byref(t), C.byref(api.f3(d)), None).tuple() for d in dirs], np.float32)
    lib.texture_destroy(C.byref(t))
    return out


def test_latlong_lookup_oracle_vs_reference(H, ref_lib, orc_lib):
    r = H.rng(22)
    dirs = np.concatenate([r.normal(size=(6000, 3)) * r.uniform(0.1, 5, size=(6000, 1)),
                           [[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1], [-1, 0, -1e-9], [-1, 0, 1e-9], [1e-4, -1, 0], [3, 4, 0]]]).astype(np.float32)
    for td in env_textures(H):
        want = latlong_all(ref_lib, td, dirs)
        for mode in (0, 1):         # libm and the device-twin math: both must give the reference's texel
            H.set_oracle_math(mode)
            try:
                got = latlong_all(orc_lib, td, dirs)
            finally:
                H.set_oracle_math(0)
            assert np.array_equal(H.bits(got), H.bits(want)), mode


def test_oracle_environment_lighting_semantics(H, orc_lib):
    U = H.Unit("orc")
    env = (0.4, 0.52, 1.0)
    # nothing to hit: every sample adds exactly the environment, so the mean is the environment itself
    d = scenes.SceneDesc(objects=[], width=24, height=16, spp=4, environment=env, environment_lighting=True)
    img = U.render_pixels(d, want_calls=False)["pixels"]
    assert np.array_equal(img, np.broadcast_to(np.array(env, np.float32), img.shape))
    # switched off (the reference's behaviour) the same scene is black
    d.environment_lighting = False
    assert not U.render_pixels(d, want_calls=False)["pixels"].any()
    # open Cornell box: the term only adds light, and only through rays that leave the box
    on = U.render_pixels(scenes.cornell_box(48, 48, 8, environment=env, environment_lighting=True))
    off = U.render_pixels(scenes.cornell_box(48, 48, 8, environment=env))
    assert np.array_equal(on["rand_calls"], off["rand_calls"])          # same paths, same draws
    assert (on["pixels"] >= off["pixels"]).all() and (on["pixels"] > off["pixels"]).mean() > 0.5
    # lat-long map, nothing to hit, no jitter: each pixel is one texel of the map
    tex = env_textures(H)[0]
    d = scenes.SceneDesc(objects=[], width=32, height=24, spp=1, jitter=0.0, environment_texture=tex, environment_lighting=True)
    img = U.render_pixels(d, want_calls=False)["pixels"].reshape(-1, 3)
    texels = {tuple(t) for t in tex.data.reshape(-1, 3)}
    assert all(tuple(p) in texels for p in img) and len({tuple(p) for p in img}) > 3


def test_reference_has_no_such_switch(H, ref_lib):
    with pytest.raises(ValueError):
        scenes.build_scene(ref_lib, scenes.cornell_box(8, 8, 1, environment_lighting=True))


# ---------------------------------------------------------------------------------------------- device


def dev_vs_oracle(H, L, d):
    from test_gpu_render import render_dev, same
    H.set_oracle_math(1)
    try:
        want = H.Unit("orc").render_pixels(d)
    finally:
        H.set_oracle_math(0)
    got = render_dev(L, d, calls=True)
    assert same(H, got["pixels"], want["pixels"]) and same(H, got["acc"], want["acc"])
    assert np.array_equal(got["rand_calls"], want["rand_calls"])
    return got


@pytest.mark.gpu
def test_device_environment_lighting_equals_oracle(H, amd_lib, orc_lib):
    from terra_amd import runtime
    L = runtime.load()
    texf, tex8 = env_textures(H)
    env = (0.4, 0.52, 1.0)
    dev_vs_oracle(H, L, scenes.cornell_box(96, 64, 4, environment=env, environment_lighting=True))
    dev_vs_oracle(H, L, scenes.cornell_phong(64, 48, 4, integrator=api.kTerraIntegratorDirect, environment_texture=texf, environment_lighting=True))
    dev_vs_oracle(H, L, scenes.cornell_box(64, 48, 3, integrator=api.kTerraIntegratorDirectMis, environment_texture=tex8, environment_lighting=True, tonemap=api.kTerraTonemappingOperatorReinhard))
    dev_vs_oracle(H, L, scenes.cornell_textured(64, 48, 2, environment_texture=texf, environment_lighting=True))
    got = dev_vs_oracle(H, L, scenes.SceneDesc(objects=[], width=40, height=24, spp=2, environment_texture=texf, environment_lighting=True))
    assert got["pixels"].any()


@pytest.mark.gpu
def test_device_environment_switch_and_rebinding(H, amd_lib):
    from terra_amd import runtime
    from test_gpu_render import render_dev
    L = runtime.load()
    env = (0.4, 0.52, 1.0)
    d = scenes.cornell_box(64, 64, 2, environment=env)
    off = render_dev(L, d)["pixels"]
    # one scene: off -> on -> other environment -> off, committing in between
    scene = scenes.build_scene(L, d)
    fb = runtime.DeviceFramebuffer(d.width, d.height); cam = scenes.camera_of(d)

    def shot():
        fb.clear(); runtime.render_device(L, cam, scene, fb); return fb.pixels_host().copy()
    assert np.array_equal(shot(), off) and L.get_environment_lighting(scene) == 0
    assert L.set_environment_lighting(scene, 1) == 0
    L.clear_error()
    rc = L.render_device(C.byref(cam), scene, fb.pixels.data_ptr(), fb.results.data_ptr(), d.width, d.height, 0, 0, d.width, d.height, None, None)
    assert rc == -2, "a changed switch needs a commit"       # kTerraAmdErrNotCommitted
    L.scene_commit(scene)
    on = shot()
    assert (on >= off).all() and (on > off).any()
    o = L.scene_get_options(scene).contents
    o.environment_map = api.const_attribute(L, (2.0, 0.0, 0.0))
    L.scene_commit(scene)
    red = shot()
    assert not np.array_equal(red, on) and (red[..., 0] >= off[..., 0]).all()
    L.set_environment_lighting(scene, 0); L.scene_commit(scene)
    assert np.array_equal(shot(), off)
    # a texture lookup by texcoord makes no sense for a direction: refused at commit, with the reason
    L.set_environment_lighting(scene, 1)
    o = L.scene_get_options(scene).contents
    o.environment_map = scenes.texture_attribute(L, env_textures(H)[0])
    L.clear_error(); L.scene_commit(scene)
    assert "environment" in runtime.last_error()
    L.clear_error()
    L.scene_destroy(scene)
