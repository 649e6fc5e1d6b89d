"""Host logic of the product that runs without a GPU: BVH build in the reference's
tree layout, scene bookkeeping, error behaviour (fail loudly, never fall back),
framebuffer/texture API, shard helpers."""
import ctypes as C

import numpy as np
import pytest

from terra_amd import api, runtime, scenes


@pytest.fixture(scope="module")
def L(amd_lib):
    return runtime.load(need_torch=False)


def test_host_bvh_equals_golden_and_oracle(H, L, orc_lib):
    d = scenes.cornell_box(256, 256, 4)
    scene = scenes.build_scene(L, d)
    nodes = H.Unit("amd").bvh_nodes(scene)
    assert np.array_equal(nodes, np.load(H.GOLDEN / "bvh_cornell.npz")["nodes"])
    info = runtime.SceneInfo()
    assert L.scene_info(scene, C.byref(info)) == 0
    assert (info.triangles, info.nodes, info.objects, info.lights, info.lights_triangles_count) == (32, 31, 6, 1, 2)
    so = scenes.build_scene(orc_lib, d)
    assert info.max_stack == orc_lib.fn("orc_bvh_max_stack", C.c_int, [C.c_void_p])(so)
    L.scene_destroy(scene)


@pytest.mark.parametrize("n_tris,seed", [(1, 1), (2, 2), (5, 3), (64, 4), (700, 5), (5000, 6)])
def test_host_bvh_equals_oracle_on_random_soups(H, L, orc_lib, n_tris, seed):
    from test_oracle_vs_reference import soup_scene
    d = soup_scene(H, n_tris, seed, n_objects=min(3, n_tris))
    sa, so = scenes.build_scene(L, d), scenes.build_scene(orc_lib, d)
    assert np.array_equal(H.Unit("amd").bvh_nodes(sa), H.Unit("orc").bvh_nodes(so))
    L.scene_destroy(sa); orc_lib.scene_destroy(so)


def test_no_device_fails_loudly_and_renders_nothing(H, L):
    if L.device_count() > 0:
        pytest.skip("a GPU is visible: covered by the gpu tests")
    d = scenes.cornell_box(16, 16, 1)
    L.clear_error()
    scene = scenes.build_scene(L, d)
    assert "no HIP device" in runtime.last_error()
    fb = api.Framebuffer(L, 16, 16)
    cam = scenes.camera_of(d)
    L.clear_error()
    L.render(C.byref(cam), scene, C.byref(fb.fb), 0, 0, 16, 16)
    assert "no device replica" in runtime.last_error()
    assert not fb.pixels.any() and not fb.results["samples"].any()      # nothing rendered, no CPU fallback
    rc = L.render_device(C.byref(cam), scene, None, None, 16, 16, 0, 0, 16, 16, None, None)
    assert rc == -1
    L.scene_destroy(scene)


def test_render_before_commit_is_an_error(L):
    scene = L.scene_create()
    L.scene_add_object(scene, 1)
    fb = api.Framebuffer(L, 8, 8)
    cam = scenes.camera_of(scenes.cornell_box())
    L.clear_error()
    L.render(C.byref(cam), scene, C.byref(fb.fb), 0, 0, 8, 8)
    assert "commit" in runtime.last_error()
    L.scene_destroy(scene)


def test_foreign_bsdf_pointers_are_rejected_at_commit(H, L):
    d = scenes.cornell_box(8, 8, 1)
    scene = L.scene_create()
    for od in d.objects:
        scenes.fill_object(L, L.scene_add_object(scene, len(od.triangles)).contents, od)
    # a client-supplied BSDF callback (allowed by the reference API) cannot run on the device
    cb = L.malloc      # any function address that is not one of the library's preset entry points
    objs = L.scene_add_object(scene, 1).contents
    scenes.fill_object(L, objs, scenes.ObjectDesc(d.objects[0].triangles[:1], d.objects[0].normals[:1], d.objects[0].texcoords[:1]))
    objs.material.bsdf.sample = C.cast(cb, C.c_void_p)
    scenes.apply_options(L, scene, d)
    L.clear_error()
    L.scene_commit(scene)
    assert "BSDF function pointers" in runtime.last_error()
    L.scene_destroy(scene)


def test_preset_markers_refuse_host_calls(L):
    b = api.TerraBSDF(); L.bsdf_diffuse_init(C.byref(b))
    f = C.CFUNCTYPE(api.TerraFloat3, C.c_void_p, C.c_void_p, C.c_void_p)(b.eval)
    L.clear_error()
    r = f(None, None, None)
    assert r.tuple() == (0.0, 0.0, 0.0) and "device only" in runtime.last_error()
    b2 = api.TerraBSDF(); L.bsdf_phong_init(C.byref(b2))
    assert b.sample != b2.sample and b.pdf != b2.pdf and b.eval != b2.eval


def test_framebuffer_api(L):
    fb = api.TerraFramebuffer()
    assert not L.framebuffer_create(C.byref(fb), 0, 4)
    f = api.Framebuffer(L, 7, 5)
    assert f.pixels.shape == (5, 7, 3) and not f.pixels.any() and not f.results["samples"].any()
    f.pixels[:] = 1; f.results["samples"][:] = 3
    f.clear()
    assert not f.pixels.any() and not f.results["samples"].any()
    f.destroy()


def test_options_are_double_buffered(L):
    scene = L.scene_create()
    o = L.scene_get_options(scene).contents
    o.samples_per_pixel = 9
    assert L.scene_get_options(scene).contents.samples_per_pixel == 9
    assert L.scene_count_objects(scene) == 0
    L.scene_add_object(scene, 2)
    assert L.scene_count_objects(scene) == 1
    L.scene_clear(scene)
    assert L.scene_count_objects(scene) == 0
    L.scene_destroy(scene)


def test_texture_api_matches_oracle(H, L, orc_lib):
    r = H.rng(3)
    data8 = r.randint(0, 256, size=(5, 4, 3)).astype(np.uint8)
    dataf = r.uniform(0, 2, size=(5, 4, 3)).astype(np.float32)
    for lib in (L, orc_lib):
        lib._tex = []
    outs = []
    for lib in (L, orc_lib):
        res = []
        for depth, data, init in ((1, data8, lib.texture_init), (4, dataf, lib.texture_init_hdr)):
            for addr in (api.kTerraAcceleratorBVH, 1, 2):
                for flt in (0, 1):
                    t = api.TerraTexture()
                    init(C.byref(t), 4, 5, 3, data.ctypes.data)
                    t.address_mode = addr; t.filter = flt
                    lib.texture_finalize(C.byref(t))
                    for (u, v) in [(0.0, 0.0), (1.5, 2.25), (3.9, 4.9), (2.0, 1.0)]:
                        uv = api.TerraFloat2(u, v)
                        res.append(lib.texture_sample(C.byref(t), C.byref(uv), None).tuple())
                    res.append(lib.texture_read(C.byref(t), 3, 4).tuple())
                    dirv = api.TerraFloat3(0.3, 0.5, -0.8)
                    res.append(lib.texture_sample_latlong(C.byref(t), C.byref(dirv), None).tuple())
                    lib.texture_destroy(C.byref(t))
        outs.append(np.array(res, np.float32))
    assert np.array_equal(H.bits(outs[0]), H.bits(outs[1]))


def test_shard_helpers(L):
    # 1080p in 128-px tiles: 15 x 9 = 135 tiles
    counts = [L.shard_tile_count(1920, 1080, 128, r, 8) for r in range(8)]
    assert sum(counts) == 135 and max(counts) - min(counts) <= 1
    assert counts == [len(runtime.shard_tiles(1920, 1080, 128, r, 8)) for r in range(8)]
    assert L.shard_packed_bytes(1920, 1080, 128, 8) == counts[0] * 128 * 128 * 28
    assert runtime.packed_floats_per_rank(1920, 1080, 128, 8) * 4 == L.shard_packed_bytes(1920, 1080, 128, 8)
    assert L.shard_tile_count(64, 64, 0, 0, 1) < 0


def test_automatic_sample_split_rule(L):
    # what terra_amd_set_sample_split(scene, 0) chooses (a pure function of the launch's size): launches that use the job order aim at ~50 jobs per resident lane ...
    assert [L.auto_sample_split(1920, 1080, 64, n, 512, 1) for n in (1, 2, 4, 8, 16)] == [8, 16, 32, 32, 32]
    # ... the others at ~200 (chunks of at least 16 samples, at most 32 lanes per pixel) ...
    assert [L.auto_sample_split(1920, 1080, 64, n, 512, 0) for n in (1, 2, 8)] == [32, 32, 32] and L.auto_sample_split(1920, 1080, 64, 1, 256, 0) == 16
    assert L.auto_sample_split(3840, 2160, 64, 1, 4096, 0) == 8 and L.auto_sample_split(1920, 1080, 64, 1, 16, 1) == 1 and L.auto_sample_split(1920, 1080, 64, 1, 48, 1) == 2
    # ... a launch below the job order's size limit is not ordered whatever the flag says (a 128-pixel tile: 64 pixel blocks)
    assert L.auto_sample_split(128, 128, 64, 1, 512, 1) == L.auto_sample_split(128, 128, 64, 1, 512, 0) == 32
    # the split always divides spp
    assert L.auto_sample_split(1920, 1080, 64, 1, 96, 0) == 4 and 96 % L.auto_sample_split(1920, 1080, 64, 1, 96, 1) == 0 and L.auto_sample_split(1920, 1080, 64, 1, 17 * 32, 0) == 32
    assert L.auto_sample_split(0, 1080, 64, 1, 512, 1) < 0 and L.auto_sample_split(1920, 1080, 24, 1, 512, 1) < 0 and L.auto_sample_split(1920, 1080, 64, 0, 512, 1) < 0
    L.clear_error()


# ---------------------------------------------------------------------------
# automatic traversal policy: the commit-time containment check (host only)
# ---------------------------------------------------------------------------

def _decision(L, d, mode=None):
    L.clear_error()
    scene = scenes.build_scene(L, d, tree_mode=mode)
    ti = runtime.TraversalInfo()
    assert L.traversal_info(scene, C.byref(ti)) == 0
    out = (ti.tree_mode, ti.fast_tree, ti.leaf_cull, ti.max_coordinate, ti.note.decode())
    L.scene_destroy(scene); L.clear_error()
    return out


def test_automatic_traversal_decision(H, L):
    from test_oracle_vs_reference import soup_scene
    # default = automatic; the LDS-resident Cornell box takes the leaf-box cull, the 97k-triangle hall the fast tree
    mode, fast, cull, cmax, note = _decision(L, scenes.cornell_box(32, 32, 1))
    assert (mode, fast, cull) == (2, 0, 1) and cmax == 2.0 and "leaf-box cull" in note
    mode, fast, cull, cmax, note = _decision(L, scenes.sponza_hall(32, 18, 1))
    assert (mode, fast, cull) == (2, 1, 0) and cmax <= 13.0 and "fast tree" in note
    # explicit modes are obeyed without a check
    assert _decision(L, scenes.cornell_box(32, 32, 1), 0)[:3] == (0, 0, 0)
    assert _decision(L, scenes.cornell_box(32, 32, 1), 1)[:3] == (1, 1, 0)
    # coordinates beyond the range in which the 1e-4 box margin provably exceeds rounding error: a scene that is not LDS-resident keeps the fast tree with the
    # reference's reachability replayed per accepted hit; a resident one keeps the reference tree and culls with LEAF boxes rebuilt at the scene's rounding bound --
    # with the reason either way
    d = soup_scene(H, 300, 9)
    for o in d.objects:
        o.triangles = (o.triangles * np.float32(40.0)).astype(np.float32)
    mode, fast, cull, cmax, note = _decision(L, d)
    assert (fast, cull) == (1, 0) and cmax > 13.0 and "exceeds" in note and "reachability" in note
    d = scenes.cornell_box(32, 32, 1)
    for o in d.objects:
        o.triangles = (np.asarray(o.triangles, np.float32) * np.float32(40.0)).astype(np.float32)
    mode, fast, cull, cmax, note = _decision(L, d)
    assert (fast, cull) == (0, 1) and cmax > 13.0 and "exceeds" in note and "leaf boxes rebuilt" in note
    # the test hook that shrinks the device's copy of the reference boxes voids the in-range containment proof: no shortcut that rests on it
    sc = scenes.build_scene(L, scenes.cornell_box(32, 32, 1), debug_shrink=0.01)
    ti = runtime.TraversalInfo(); assert L.traversal_info(sc, C.byref(ti)) == 0
    assert (ti.fast_tree, ti.leaf_cull) == (0, 0) and "TEST HOOK" in ti.note.decode()
    L.scene_destroy(sc)
    # non-finite coordinates never pass
    d = soup_scene(H, 50, 10); d.objects[0].triangles[0, 0, 0] = np.float32("nan")
    assert _decision(L, d)[1:3] == (0, 0)
    # fewer than 2 triangles: nothing to cull
    d = soup_scene(H, 1, 11, n_objects=1)
    assert _decision(L, d)[1:3] == (0, 0)


def test_fast_tree_too_deep_for_the_lds_stack_falls_back(H, L):
    """The fast-tree launch plans 1 KB of LDS per stack entry and block (terra_plan_fast_tree); a tree whose depth would exceed what a block may ask for must not
    be chosen at commit -- the scene keeps the reference tree, which always fits -- instead of failing every render. Clustered geometry does it: 5,000 triangles
    whose extents nest like onion shells force a chain-like tree on any builder."""
    n = 5000
    k = np.arange(n, dtype=np.float64)
    r = (1e-3 * 1.002 ** k).astype(np.float32)                      # nested shells from 1e-3 to ~22 units: every split separates one shell from the rest
    tris = np.zeros((n, 3, 3), np.float32)
    tris[:, 0, 0] = r; tris[:, 1, 1] = r; tris[:, 2, 2] = r          # triangle k spans (r,0,0) (0,r,0) (0,0,r): box [0, r]^3, all sharing the origin corner
    nrm = np.tile(np.array([0.577, 0.577, 0.577], np.float32), (n, 3, 1))
    od = scenes.ObjectDesc(triangles=tris, normals=nrm, texcoords=np.zeros((n, 3, 2), np.float32), material=scenes.Material(kind="diffuse", albedo=(0.5, 0.5, 0.5)))
    d = scenes.SceneDesc(objects=[od], width=16, height=16, spp=1)
    for mode in (2, 1):
        L.clear_error()
        scene = scenes.build_scene(L, d, tree_mode=mode)
        ti = runtime.TraversalInfo(); assert L.traversal_info(scene, C.byref(ti)) == 0
        note = ti.note.decode()
        if ti.fast_tree:                                                 # shallow enough after all: then it must fit the limit the commit checks
            assert "stack" not in note
        else:
            assert "stack" in note and "reference tree" in note, note
        L.scene_destroy(scene); L.clear_error()


def test_first_error_is_process_wide_and_sticky(H, L):
    """a worker thread's failure is visible to the thread that polls; the first one is kept until cleared"""
    import threading
    L.clear_first_error(); L.clear_error()
    assert runtime.first_error() == (0, "")
    def worker():
        L.set_tree_mode(L.scene_create(), 7)           # an error on another thread
    t = threading.Thread(target=worker); t.start(); t.join()
    assert runtime.last_error() == ""                   # this thread's channel is untouched ...
    st, msg = runtime.first_error()
    assert st < 0 and "tree mode 7" in msg              # ... the process-wide one has it
    L.set_sample_split(L.scene_create(), 3)             # a later error does not replace the first
    assert "sample split" in runtime.last_error() and "tree mode 7" in runtime.first_error()[1]
    L.clear_first_error(); L.clear_error()
    assert runtime.first_error() == (0, "")


def test_binary16_planes_are_rounded_outward(H, L):
    """the fast tree's node planes (tree_build.cpp fastbvh::half_outward): the result is the TIGHTEST binary16 on the wanted side of x -- checked against numpy's float16 for
    normal, subnormal, tiny, huge and special values; a box made of such planes contains the box it was made from"""
    f = L.fn("terra_amd_unit_half_outward", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p])
    r = H.rng(5)
    x = np.concatenate([r.uniform(-14, 14, 20000), r.uniform(-1, 1, 5000) * 10.0 ** r.uniform(-12, 5, 5000), np.float64(np.arange(-2100, 2100).astype(np.float16)) * 1.0,
                        [0.0, -0.0, 65504.0, 65504.1, 65519.9, 65520.0, 1e6, -65504.0, -65504.1, -1e6, 5.96e-8, 2.9e-8, 1e-30, -1e-30, 6.1e-5, -6.1e-5, np.inf, -np.inf, 1e-320]]).astype(np.float64)
    for up in (0, 1):
        out = np.zeros(len(x), np.uint16)
        assert f(x.ctypes.data, len(x), up, out.ctypes.data) == 0
        h = out.view(np.float16).astype(np.float64)
        with np.errstate(over="ignore"):
            if up:
                assert (h >= x).all()
                tighter = np.nextafter(out.view(np.float16), np.float16(-np.inf)).astype(np.float64)      # the next binary16 below must already be on the wrong side
                ok = (tighter < x) | (h == x)
            else:
                assert (h <= x).all()
                tighter = np.nextafter(out.view(np.float16), np.float16(np.inf)).astype(np.float64)
                ok = (tighter > x) | (h == x)
        assert ok.all(), (x[~ok][:5], h[~ok][:5])
    nan = np.array([np.nan]); o = np.zeros(1, np.uint16)
    assert f(nan.ctypes.data, 1, 1, o.ctypes.data) == 0 and o[0] == 0x7c00 and f(nan.ctypes.data, 1, 0, o.ctypes.data) == 0 and o[0] == 0xfc00      # the widest value either way


def test_scene_supported_query(H, L):
    """terra_amd_scene_supported: what the commit would decide about the scene's materials, asked without committing and without touching the error channels -- the reference
    runs any callback (src/Terra.c:1071-1075, 1804-1810), the device runs the presets and texture lookups only"""
    d = scenes.cornell_box(8, 8, 1)
    why = C.create_string_buffer(256)
    scene = L.scene_create()
    for od in d.objects:
        scenes.fill_object(L, L.scene_add_object(scene, len(od.triangles)).contents, od)
    scenes.apply_options(L, scene, d)
    L.clear_error(); L.clear_first_error()
    st = L.scene_supported(scene, why, 256)
    if L.device_count() > 0:
        assert st == 0 and why.value == b""
    else:
        assert st == -1 and b"no HIP device" in why.value                 # (the only objection on a box without a GPU)
    extra = L.scene_add_object(scene, 1).contents
    scenes.fill_object(L, extra, scenes.ObjectDesc(d.objects[0].triangles[:1], d.objects[0].normals[:1], d.objects[0].texcoords[:1]))
    extra.material.bsdf.pdf = C.cast(L.malloc, C.c_void_p)                 # a client-supplied callback
    st = L.scene_supported(scene, why, 256)
    assert st == -3 and b"object 6" in why.value and b"preset" in why.value
    L.bsdf_diffuse_init(C.byref(extra.material.bsdf))
    extra.material.attributes[0].state = C.cast(L.malloc, C.c_void_p); extra.material.attributes[0].eval = C.cast(L.malloc, C.c_void_p)      # an attribute callback
    st = L.scene_supported(scene, why, 256)
    assert st == -3 and b"attribute 0" in why.value and b"terra_texture_sample" in why.value
    assert L.scene_supported(scene, None, 0) == -3                           # the reason is optional
    assert runtime.last_error() == "" and runtime.first_error() == (0, "")   # a pure query: nothing recorded
    L.scene_destroy(scene)
