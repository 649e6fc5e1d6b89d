"""SURVEY.md 8f N4, the wiring: the reference constructs a stratified or Halton "hemisphere sampler" per pixel (src/Terra.c:535-548)
and never draws from it. Behind terra_amd_set_sampler_integration(scene, 1) -- off by default, mirrored by the oracle's
orc_set_sampler_integration -- camera sample n of a pixel takes element n of that sampler (Halton: the radical-inverse pair of n,
src/Terra.c:734-755; stratified: the sampler of :542, 16 samples per stratum, offsets from the pixel's camera stream, :714-723) and uses
it as the first two variates of the BSDF sample at bounce 0. PARITY UNPINNED: the wiring is this repo's definition, there is nothing in
the reference to pin it to; what is pinned is device == oracle bit for bit, and that with the switch off nothing changes (every golden
test runs with it off)."""
import ctypes as C

import numpy as np
import pytest

from terra_amd import api, runtime, scenes

METHODS = {"halton": api.kTerraSamplingMethodHalton, "stratified": api.kTerraSamplingMethodStratified}


def desc(method, integ, spp=16, on=True, w=72, h=48):
    d = scenes.cornell_box(w, h, spp, integrator=integ)
    d.sampling = METHODS.get(method, api.kTerraSamplingMethodRandom); d.strata = 2          # stratified: 2 x 2 strata x 16 = 64 elements; spp 16 = strata^(2^1)
    d.sampler_integration = on
    return d


def test_oracle_switch_changes_only_the_sampler_methods(H, orc_lib):
    """with the random sampling method the switch is inert; with Halton / stratified it moves the first bounce (and only then)"""
    U = H.Unit("orc")
    for method in ("random", "halton", "stratified"):
        off = U.render_pixels(desc(method, 0, on=False), threads=4)
        on = U.render_pixels(desc(method, 0, on=True), threads=4)
        again = U.render_pixels(desc(method, 0, on=True), threads=2)
        assert H.same_bits(on["acc"], again["acc"])                                      # deterministic, thread independent
        changed = not H.same_bits(on["acc"], off["acc"])
        assert changed == (method != "random"), method
        if method != "stratified":       # stream B is consumed exactly as without the switch; only the stratified sampler draws (from stream A)
            assert on["rand_calls"].sum() > 0
    # the pair really is the Halton element: a one-sample render with the switch on equals one whose FIRST stream-B pair is irrelevant ... checked on the device side below


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["halton", "stratified"])
@pytest.mark.parametrize("integ", [0, 1, 2])
def test_device_equals_oracle_with_the_switch_on(H, amd_lib, orc_lib, devmath_mode, method, integ):
    import torch
    L = runtime.load()
    d = desc(method, integ)
    want = H.Unit("orc").render_pixels(d, passes=2, threads=8)           # two progressive passes: the sampler continues at element (samples already in the pixel)
    scene = scenes.build_scene(L, d)
    fb = runtime.DeviceFramebuffer(d.width, d.height); cam = scenes.camera_of(d)
    rc = torch.zeros(d.width * d.height, dtype=torch.int32, device="cuda")
    for _ in range(2):
        runtime.render_device(L, cam, scene, fb, None, rc)
    torch.cuda.synchronize()
    res = fb.results_host()
    assert H.same_bits(res["acc"], want["acc"]) and H.same_bits(fb.pixels_host(), want["pixels"]) and np.array_equal(res["samples"], want["samples"])
    assert np.array_equal(rc.cpu().numpy().reshape(d.height, d.width).astype(np.uint64), want["rand_calls"].astype(np.uint64))
    # a sample split is the frame of that many successive calls: chunk j starts at its own element
    fb2 = runtime.DeviceFramebuffer(d.width, d.height)
    runtime.check(L.set_sample_split(scene, 4))
    runtime.render_device(L, cam, scene, fb2); torch.cuda.synchronize()
    d4 = desc(method, integ, spp=4)
    if method == "halton":               # (stratified: spp 4 would be rounded up to strata^2 = 4 -- the same -- but its streams differ per pass anyway; Halton pins the indexing)
        want4 = H.Unit("orc").render_pixels(d4, passes=4, threads=8)
        assert H.same_bits(fb2.results_host()["acc"], want4["acc"])
    # switch off on the same scene handle: the plain image again (a launch parameter, no commit needed)
    runtime.check(L.set_sampler_integration(scene, 0)); runtime.check(L.set_sample_split(scene, 1))
    fb3 = runtime.DeviceFramebuffer(d.width, d.height)
    runtime.render_device(L, cam, scene, fb3); torch.cuda.synchronize()
    off = H.Unit("orc").render_pixels(desc(method, integ, on=False), threads=8)
    assert H.same_bits(fb3.results_host()["acc"], off["acc"])
    L.scene_destroy(scene)
