"""SURVEY.md 8f N4, second half: "TerraDistribution1D/2D for env-map importance sampling". Nothing in the reference calls its
TerraDistribution2D (src/Terra.c:812-846), so this wiring is this repo's definition -- PARITY UNPINNED -- behind
terra_amd_set_environment_sampling(scene, 1), off by default, mirrored by the oracle's orc_set_environment_sampling: with environment
lighting on and a lat-long environment texture, the Direct and Direct+MIS integrators take one environment sample per shaded hit (two draws of
stream B pick a texel through a TerraDistribution2D over luminance x sin(theta); a shadow ray; radiance of that texel x BSDF x cosine / density)
and a path ray that leaves the scene after bounce 0 no longer adds the environment. What is checked: the table and its sampling are the
pinned unit-level functions (tests/test_samplers.py); the estimator agrees in the mean with the plain one (it is the same integral); device ==
oracle bit for bit; with the switch off nothing changes (every golden test runs with it off)."""
import numpy as np
import pytest

from terra_amd import api, runtime, scenes


def sky(h=16, w=32, sun=60.0):
    """a dim sky with a small bright patch above the horizon: most of the light comes from 4 of 512 texels"""
    t = np.full((h, w, 3), 0.15, np.float32)
    t[: h // 2] += np.float32(0.1)
    t[3:5, 9:11] = (sun, sun * 0.9, sun * 0.7)
    return scenes.TextureDesc(t)


def courtyard(width, height, spp, integrator, sampling, tex=None, bounces=3, **kw):
    """open scene: floor, one box, a small emissive panel (the light integrators need one emissive object), lit by the sky"""
    floor = scenes._merge([scenes._quad((-2, 0, -2), (2, 0, -2), (2, 0, 2), (-2, 0, 2), (0, 1, 0))])
    box = scenes._open_box(-0.4, 0.4, 0.0, 0.8, -0.2, 0.6)
    panel = scenes._merge([scenes._quad((-0.2, 1.6, 0.8), (0.2, 1.6, 0.8), (0.2, 1.6, 1.2), (-0.2, 1.6, 1.2), (0, -1, 0))])
    objs = [scenes.ObjectDesc(*floor, scenes.Material(albedo=(0.7, 0.7, 0.7)), "floor"),
            scenes.ObjectDesc(*box, scenes.Material(albedo=(0.6, 0.3, 0.2)), "box"),
            scenes.ObjectDesc(*panel, scenes.Material(albedo=(0.5, 0.5, 0.5), emissive=(4.0, 4.0, 4.0)), "panel")]
    d = scenes.SceneDesc(objects=objs, width=width, height=height, spp=spp, bounces=bounces, integrator=integrator, name="courtyard",
                         environment_texture=tex if tex is not None else sky(), environment_lighting=True, environment_sampling=sampling)
    d.camera_position = (0.0, 1.2, -3.0); d.camera_direction = (0.0, -0.25, 1.0)
    for k, v in kw.items():
        setattr(d, k, v)
    return d


def test_switch_is_inert_where_it_does_not_apply(H, orc_lib):
    U = H.Unit("orc")
    # Simple integrator: no light samples at all
    a = U.render_pixels(courtyard(24, 16, 4, api.kTerraIntegratorSimple, False), threads=4)
    b = U.render_pixels(courtyard(24, 16, 4, api.kTerraIntegratorSimple, True), threads=4)
    assert H.same_bits(a["acc"], b["acc"]) and np.array_equal(a["rand_calls"], b["rand_calls"])
    # constant environment: there is no map to tabulate
    c0 = courtyard(24, 16, 4, api.kTerraIntegratorDirect, False); c0.environment_texture = None; c0.environment = (0.3, 0.4, 0.5)
    c1 = courtyard(24, 16, 4, api.kTerraIntegratorDirect, True); c1.environment_texture = None; c1.environment = (0.3, 0.4, 0.5)
    a = U.render_pixels(c0, threads=4); b = U.render_pixels(c1, threads=4)
    assert H.same_bits(a["acc"], b["acc"])
    # environment lighting off: the switch alone does nothing
    c0 = courtyard(24, 16, 4, api.kTerraIntegratorDirect, False); c0.environment_lighting = False
    c1 = courtyard(24, 16, 4, api.kTerraIntegratorDirect, True); c1.environment_lighting = False
    a = U.render_pixels(c0, threads=4); b = U.render_pixels(c1, threads=4)
    assert H.same_bits(a["acc"], b["acc"])
    # where it applies: two more draws per shaded hit, another image, deterministic and thread independent
    a = U.render_pixels(courtyard(24, 16, 4, api.kTerraIntegratorDirect, False), threads=4)
    b = U.render_pixels(courtyard(24, 16, 4, api.kTerraIntegratorDirect, True), threads=4)
    b2 = U.render_pixels(courtyard(24, 16, 4, api.kTerraIntegratorDirect, True), threads=2)
    assert not H.same_bits(a["acc"], b["acc"]) and H.same_bits(b["acc"], b2["acc"]) and b["rand_calls"].sum() > a["rand_calls"].sum()


@pytest.mark.parametrize("integ", [api.kTerraIntegratorDirect, api.kTerraIntegratorDirectMis])
def test_the_sampled_estimator_has_the_plain_estimators_mean_and_less_noise(H, orc_lib, integ):
    """same integral, two estimators: path rays that happen to escape into the sky (plain) against one sky sample per hit. Means over the
    image agree within the noise; the per-pixel spread of the sampled one is far smaller (the sun patch is 0.8 % of the map)."""
    U = H.Unit("orc")
    plain = U.render_pixels(courtyard(32, 24, 4096, integ, False), threads=8)
    samp = U.render_pixels(courtyard(32, 24, 512, integ, True), threads=8)
    mp = (plain["acc"] / plain["samples"][..., None]).astype(np.float64); ms = (samp["acc"] / samp["samples"][..., None]).astype(np.float64)
    assert abs(ms.mean() / mp.mean() - 1) < 0.02, (ms.mean(), mp.mean())
    # rows of the image (floor far / box / floor near) agree one by one as well
    for band in np.array_split(np.arange(24), 4):
        assert abs(ms[band].mean() / mp[band].mean() - 1) < 0.04
    # noise: two independent halves of each estimator at EQUAL sample counts
    def halves(sampling, seed):
        return [U.render_pixels(courtyard(32, 24, 64, integ, sampling), frame_seed=seed + k, threads=8) for k in range(2)]
    def spread(pair):
        a, b = [(r["acc"] / r["samples"][..., None]).astype(np.float64) for r in pair]
        return np.sqrt(((a - b) ** 2).mean())
    assert spread(halves(True, 100)) < 0.5 * spread(halves(False, 200))


@pytest.mark.gpu
@pytest.mark.parametrize("integ", [api.kTerraIntegratorDirect, api.kTerraIntegratorDirectMis])
def test_device_equals_oracle_with_environment_sampling(H, amd_lib, orc_lib, devmath_mode, integ):
    import torch
    L = runtime.load()
    for tex in (sky(), scenes.TextureDesc(H.rng(5).randint(0, 256, size=(7, 13, 3)).astype(np.uint8), address_mode=2)):
        d = courtyard(72, 48, 8, integ, True, tex=tex)
        want = H.Unit("orc").render_pixels(d, passes=2, threads=8)
        scene = scenes.build_scene(L, d)
        fb = runtime.DeviceFramebuffer(d.width, d.height); cam = scenes.camera_of(d)
        rc = torch.zeros(d.width * d.height, dtype=torch.int32, device="cuda")
        for _ in range(2):
            runtime.render_device(L, cam, scene, fb, None, rc)
        torch.cuda.synchronize()
        res = fb.results_host()
        assert H.same_bits(res["acc"], want["acc"]) and H.same_bits(fb.pixels_host(), want["pixels"])
        assert np.array_equal(rc.cpu().numpy().reshape(d.height, d.width).astype(np.uint64), want["rand_calls"].astype(np.uint64))
        # a sample split is the frame of that many successive calls
        runtime.check(L.set_sample_split(scene, 4))
        fb2 = runtime.DeviceFramebuffer(d.width, d.height)
        runtime.render_device(L, cam, scene, fb2); torch.cuda.synchronize()
        want4 = H.Unit("orc").render_pixels(courtyard(72, 48, 2, integ, True, tex=tex), passes=4, threads=8)
        assert H.same_bits(fb2.results_host()["acc"], want4["acc"])
        # switching it off on the same handle needs a commit (the table belongs to the committed scene) and gives the plain image
        runtime.check(L.set_environment_sampling(scene, 0)); runtime.check(L.set_sample_split(scene, 1)); L.scene_commit(scene)
        fb3 = runtime.DeviceFramebuffer(d.width, d.height)
        runtime.render_device(L, cam, scene, fb3); torch.cuda.synchronize()
        off = H.Unit("orc").render_pixels(courtyard(72, 48, 8, integ, False, tex=tex), threads=8)
        assert H.same_bits(fb3.results_host()["acc"], off["acc"])
        L.scene_destroy(scene)


@pytest.mark.gpu
def test_device_environment_sampling_in_a_closed_room_and_with_all_presets(H, amd_lib, orc_lib, devmath_mode):
    """the Cornell box sees the sky through its open side only; the sphere scene adds glass and GGX lobes to the BSDF evaluation"""
    from test_environment import dev_vs_oracle
    L = runtime.load()
    dev_vs_oracle(H, L, scenes.cornell_box(64, 48, 4, integrator=api.kTerraIntegratorDirect, environment_texture=sky(), environment_lighting=True, environment_sampling=True))
    dev_vs_oracle(H, L, scenes.cornell_spheres(64, 48, 4, integrator=api.kTerraIntegratorDirectMis, environment_texture=sky(), environment_lighting=True, environment_sampling=True))
    dev_vs_oracle(H, L, scenes.cornell_textured(48, 32, 2, integrator=api.kTerraIntegratorDirect, environment_texture=sky(8, 8), environment_lighting=True, environment_sampling=True))
