"""GGX conductor and dielectric glass (BASELINE.json config 4). The reference has no runnable
form of either (dead code), so there is nothing to pin them to: "parity
unpinned". These tests check the oracle's definitions for physical sanity on the CPU and
(gpu) that the device reproduces the oracle bit for bit."""
import numpy as np
import pytest

from terra_amd import api, scenes


def test_glass_obeys_snell_and_weights_paths_by_the_tint(H, orc_lib):
    surf, e, wo = H.bsdf_cases(201, 4000, 3)
    wi, pdf, f, after = H.Unit("orc").bsdf(3, surf, e, wo)
    n = surf[:, 16:19]; ior = surf[:, 22]; tint = surf[:, 23:26]
    cos_o = np.einsum("nc,nc->n", wo, n); cos_i = np.einsum("nc,nc->n", wi, n)
    assert (pdf > 0).all() and (pdf <= 1).all()
    # terra_trace's weight: eval / pdf * dot(n, wi) == tint
    w = f / pdf[:, None] * cos_i[:, None]
    assert np.allclose(w, tint, rtol=2e-6, atol=1e-7)
    assert np.allclose(np.linalg.norm(wi, axis=1), 1, atol=1e-5)
    refl = np.sign(cos_o) == np.sign(cos_i)
    trans = ~refl
    assert refl.any() and trans.any()
    # reflection: mirror about the normal
    r = 2 * cos_o[:, None] * n - wo
    assert np.allclose(wi[refl], r[refl], atol=1e-5)
    # refraction: n1 sin(theta1) = n2 sin(theta2)
    s_o = np.sqrt(np.maximum(0, 1 - cos_o ** 2)); s_i = np.sqrt(np.maximum(0, 1 - cos_i ** 2))
    n1 = np.where(cos_o > 0, 1.0, ior); n2 = np.where(cos_o > 0, ior, 1.0)
    assert np.allclose((n1 * s_o)[trans], (n2 * s_i)[trans], atol=2e-4)
    # total internal reflection never transmits
    tir = (cos_o < 0) & (ior * s_o > 1.0)
    assert tir.any() and refl[tir].all() and np.allclose(pdf[tir], 1.0)
    # a delta lobe: any other direction has zero pdf and zero value
    other = H.Unit("orc").bsdf(3, after, np.ones_like(e) * 0.5, wo)      # re-sample with e3 = 0.5 ...
    assert (other[1] > 0).all()


def test_ggx_is_a_reflection_lobe_with_bounded_energy(H, orc_lib):
    surf, e, wo = H.bsdf_cases(202, 20000, 2)
    wi, pdf, f, _ = H.Unit("orc").bsdf(2, surf, e, wo)
    n = surf[:, 16:19]
    # when the sampled half vector faces away from wo the lobe degenerates to wi = -wo (the dead code's
    # max(0, h.wo), src/TerraPresets.c:344): eval is 0 there and the path dies
    back = np.linalg.norm(wi + wo, axis=1) < 1e-6
    assert back.mean() < 0.2 and not f[back].any()
    wi, wo, n, f, pdf = wi[~back], wo[~back], n[~back], f[~back], pdf[~back]
    h = wi + wo; h /= np.linalg.norm(h, axis=1, keepdims=True)
    # wi is wo mirrored about a half vector in the upper hemisphere
    assert (np.einsum("nc,nc->n", h, n) > -1e-5).all()
    wide = np.linalg.norm(wi + wo, axis=1) > 0.2          # the half vector is ill-conditioned near wi = -wo
    assert np.allclose(np.einsum("nc,nc->n", wi, h)[wide], np.einsum("nc,nc->n", wo, h)[wide], atol=1e-4)
    assert np.isfinite(f).all() and (f >= 0).all() and (pdf >= 0).all()
    up = (np.einsum("nc,nc->n", wi, n) > 0) & (pdf > 1e-6)
    weight = f[up] * np.einsum("nc,nc->n", wi[up], n[up])[:, None] / pdf[up][:, None]
    # single-scatter microfacet model with F <= 1: the mean path weight cannot exceed 1
    assert weight.mean() < 1.0 and np.median(weight) > 0.05


@pytest.mark.gpu
@pytest.mark.parametrize("kind_id", [2, 3])
def test_device_bsdf_matches_oracle(H, orc_lib, amd_lib, kind_id):
    from terra_amd import runtime
    assert runtime.load().device_count() > 0
    surf, e, wo = H.bsdf_cases(210 + kind_id, 8192, kind_id)
    a, b = H.Unit("amd").bsdf(kind_id, surf, e, wo), H.Unit("orc").bsdf(kind_id, surf, e, wo)
    for x, y in zip(a[:3], b[:3]):
        assert np.array_equal(np.isnan(x), np.isnan(y)) and np.array_equal(H.bits(x)[~np.isnan(x)], H.bits(y)[~np.isnan(y)])
    assert H.same_bits(a[3][:, 29:33], b[3][:, 29:33])        # scratch slots (glass: chosen direction and probability)


@pytest.mark.gpu
@pytest.mark.parametrize("integ", [0, 1, 2])
def test_config4_scene_matches_oracle(H, orc_lib, amd_lib, devmath_mode, integ):
    from test_gpu_render import render_dev, render_host
    from terra_amd import runtime
    L = runtime.load()
    d = scenes.cornell_spheres(72, 48, 2, integrator=integ)
    assert d.triangle_count == 3980
    want = H.Unit("orc").render_pixels(d, passes=2)
    got = render_dev(L, d, passes=2, calls=True)
    assert np.array_equal(got["rand_calls"], want["rand_calls"])
    assert H.same_bits(got["pixels"], want["pixels"])
    assert H.same_bits(render_host(L, d, passes=2)["pixels"], want["pixels"])
    assert np.isfinite(got["pixels"]).all() and (got["pixels"] >= 0).all()
