"""GPU parity, image level: terra_render / terra_amd_render_device against the
golden framebuffers dumped from the compiled reference (bit-exact, including the
per-pixel rand-call counts), against the oracle on other sizes, and through
size-independent properties at BASELINE.json's full sizes."""
import ctypes as C
import json
import threading

import numpy as np
import pytest

from terra_amd import api, runtime, scenes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L(amd_lib):
    lib = runtime.load()
    assert lib.device_count() > 0, "gpu tests need a visible MI355X: " + runtime.last_error()
    return lib


def G(H, name):
    return np.load(H.GOLDEN / f"{name}.npz")


def render_host(L, d, passes=1, rects=None, threads=False, seed=None):
    """through the drop-in entry point terra_render() on a host framebuffer"""
    scene = scenes.build_scene(L, d)
    if seed is not None:
        L.set_frame_seed(scene, seed)
    fb = api.Framebuffer(L, d.width, d.height)
    cam = scenes.camera_of(d)
    rects = rects or [(0, 0, d.width, d.height)]
    L.clear_error()
    for _ in range(passes):
        if threads:
            ts = [threading.Thread(target=L.render, args=(C.byref(cam), scene, C.byref(fb.fb), *r)) for r in rects]
            [t.start() for t in ts]; [t.join() for t in ts]
        else:
            for r in rects:
                L.render(C.byref(cam), scene, C.byref(fb.fb), *r)
    assert runtime.last_error() == "", runtime.last_error()
    out = dict(pixels=fb.pixels.copy(), acc=fb.results["acc"].copy(), samples=fb.results["samples"].copy())
    fb.destroy(); L.scene_destroy(scene)
    return out


def render_dev(L, d, passes=1, rect=None, calls=False, shard=None, tree_mode=0, counters=True):
    """counters=False: the kernels a client gets by default (the counting kernels are separate instantiations; a per-pixel draw-count buffer -- calls=True -- turns
    counting on whatever the switch says)"""
    import torch
    scene = scenes.build_scene(L, d, tree_mode=tree_mode, counters=counters)
    fb = runtime.DeviceFramebuffer(d.width, d.height)
    cam = scenes.camera_of(d)
    rc = torch.zeros(d.width * d.height, dtype=torch.int32, device="cuda") if calls else None
    for _ in range(passes):
        if shard:
            tile, world = shard
            for rank in range(world):
                runtime.render_device_sharded(L, cam, scene, fb, tile, rank, world)
        else:
            runtime.render_device(L, cam, scene, fb, rect, rc)
    torch.cuda.synchronize()
    res = fb.results_host()
    out = dict(pixels=fb.pixels_host().copy(), acc=res["acc"].copy(), samples=res["samples"].copy())
    if calls:
        out["rand_calls"] = rc.cpu().numpy().astype(np.uint32).reshape(d.height, d.width)
    st = runtime.Stats(); runtime.check(L.get_stats(scene, C.byref(st))); out["stats"] = st.as_dict()
    L.scene_destroy(scene)
    return out


def close_counts(a, b):
    return abs(int(a) - int(b)) <= 2 + 2e-5 * max(int(a), int(b))


def same(H, a, b):
    a = np.asarray(a, np.float32); b = np.asarray(b, np.float32)
    nan = np.isnan(a)
    return np.array_equal(nan, np.isnan(b)) and np.array_equal(H.bits(a)[~nan], H.bits(b)[~nan])


# ---------------------------------------------------------------------------
# golden framebuffers (the reference's own output)
# ---------------------------------------------------------------------------

def test_config1_is_bit_identical_to_the_reference(H, L):
    """BASELINE.json configs[0]: Cornell box, 256x256, 4 spp, fixed seed"""
    g = G(H, "render_config1"); man = json.loads((H.GOLDEN / "manifest.json").read_text())
    d = scenes.cornell_box(256, 256, 4)
    host = render_host(L, d)
    assert H.same_bits(host["acc"], g["acc"]) and (host["samples"] == int(g["samples"])).all()
    assert H.digest(host["pixels"]) == man["config1"]["pixels_sha256"]
    dev = render_dev(L, d, calls=True)
    assert H.same_bits(dev["acc"], g["acc"]) and H.same_bits(dev["pixels"], host["pixels"])
    assert np.array_equal(dev["rand_calls"], g["rand_calls"].astype(np.uint32))
    rms = np.sqrt(np.mean((dev["pixels"].astype(np.float64) - host["pixels"]) ** 2, axis=(0, 1)))
    assert (rms <= 1e-4).all()          # north_star's stated tolerance; the actual difference is 0
    s = dev["stats"]
    assert s["samples"] == 256 * 256 * 4 and s["pixels"] == 256 * 256 and s["rand_calls"] == int(g["rand_calls"].astype(np.uint64).sum())


def test_small_goldens_all_integrators_tonemaps_presets(H, L):
    g = G(H, "render_small")
    for sname, mk in [("cornell", scenes.cornell_box), ("phong", scenes.cornell_phong)]:
        for integ in range(7):
            for tm in ([0, 1, 2, 3, 4] if (integ == 0 and sname == "cornell") else [0]):
                key = f"{sname}_i{integ}_t{tm}"
                d = mk(48, 32, 3, integrator=integ, tonemap=tm)
                out = render_dev(L, d, passes=2, calls=False)
                assert same(H, out["acc"], g[key + "_acc"]), key
                assert same(H, out["pixels"], g[key + "_pixels"]), key
                one = render_dev(L, mk(48, 32, 3, integrator=integ, tonemap=tm), passes=1, calls=True)
                host = render_host(L, d, passes=2)
                assert same(H, host["pixels"], g[key + "_pixels"]), key
    out = render_host(L, scenes.cornell_box(16, 16, 5, sampling=api.kTerraSamplingMethodStratified, strata=2))
    assert H.same_bits(out["pixels"], g["stratified_pixels"]) and np.array_equal(out["samples"], g["stratified_samples"])
    out = render_host(L, scenes.cornell_box(160, 90, 2), rects=[(48, 16, 64, 32)])
    assert H.same_bits(out["pixels"], g["tile_pixels"]) and np.array_equal(out["samples"], g["tile_samples"])


def test_rand_call_counts_match_golden_second_pass(H, L):
    g = G(H, "render_small")
    d = scenes.cornell_box(48, 32, 3, integrator=2)
    out = render_dev(L, d, passes=2, calls=True)      # counts of the LAST pass, as the reference harness records them
    assert np.array_equal(out["rand_calls"], g["cornell_i2_t0_calls"].astype(np.uint32))


# ---------------------------------------------------------------------------
# oracle on other shapes
# ---------------------------------------------------------------------------

@pytest.mark.parametrize("integ", [0, 1, 2])
def test_odd_sizes_vs_oracle(H, L, orc_lib, devmath_mode, integ):
    d = scenes.cornell_phong(101, 67, 2, integrator=integ, bounces=5, jitter=0.25, exposure=1.7)
    want = H.Unit("orc").render_pixels(d)
    got = render_dev(L, d, calls=True)
    assert np.array_equal(got["rand_calls"], want["rand_calls"])
    assert same(H, got["pixels"], want["pixels"]) and same(H, got["acc"], want["acc"])


def test_zero_bounces_and_zero_jitter(H, L, orc_lib, devmath_mode):
    for kw in (dict(bounces=0), dict(jitter=0.0), dict(bounces=1, jitter=1.0)):
        d = scenes.cornell_box(40, 24, 2, integrator=1, **kw)
        assert same(H, render_dev(L, d)["pixels"], H.Unit("orc").render_pixels(d)["pixels"]), kw


def test_random_soup_vs_oracle(H, L, orc_lib, devmath_mode):
    from test_oracle_vs_reference import soup_scene
    for n, integ in ((300, 2), (3000, 0)):
        d = soup_scene(H, n, 13, integrator=integ)
        d.width, d.height, d.spp = 64, 40, 2
        want = H.Unit("orc").render_pixels(d)
        got = render_dev(L, d, calls=True)
        assert np.array_equal(got["rand_calls"], want["rand_calls"])
        assert same(H, got["pixels"], want["pixels"])


# ---------------------------------------------------------------------------
# boundary behaviour of the drop-in entry point
# ---------------------------------------------------------------------------

def test_tiles_threads_and_device_entry_agree(H, L):
    d = scenes.cornell_box(128, 96, 3, integrator=1)
    whole = render_host(L, d, passes=2)
    tiles = [(x, y, 64, 32) for y in range(0, 96, 32) for x in range(0, 128, 64)]
    assert H.same_bits(render_host(L, d, passes=2, rects=tiles)["pixels"], whole["pixels"])
    assert H.same_bits(render_host(L, d, passes=2, rects=tiles, threads=True)["pixels"], whole["pixels"])
    assert H.same_bits(render_dev(L, d, passes=2)["pixels"], whole["pixels"])
    odd = [(0, 0, 37, 96), (37, 0, 91, 50), (37, 50, 91, 46)]
    assert H.same_bits(render_host(L, d, passes=2, rects=odd)["pixels"], whole["pixels"])


def test_progressive_accumulation(H, L):
    d = scenes.cornell_box(64, 48, 2)
    one, two = render_host(L, d, passes=1), render_host(L, d, passes=2)
    assert (one["samples"] == 2).all() and (two["samples"] == 4).all()
    assert not np.array_equal(one["acc"], two["acc"])                     # the second pass draws new streams (keyed by samples so far)
    assert (two["acc"] >= one["acc"]).all()
    m = two["acc"] / 4.0
    assert np.array_equal(H.bits(m.astype(np.float32)), H.bits(two["pixels"]))   # tonemap None, exposure 1: pixel = acc / samples


def test_frame_seed_and_determinism(H, L):
    d = scenes.cornell_box(64, 48, 2)
    a, b = render_host(L, d), render_host(L, d)
    assert H.same_bits(a["pixels"], b["pixels"])
    c = render_host(L, d, seed=1234)
    assert not H.same_bits(a["pixels"], c["pixels"])


def test_sharded_render_pack_gather_unpack(H, L):
    import torch
    d = scenes.cornell_box(200, 136, 2, integrator=1)
    whole = render_dev(L, d)
    for tile, world in ((64, 3), (32, 8), (16, 2)):
        assert H.same_bits(render_dev(L, d, shard=(tile, world))["pixels"], whole["pixels"])
    # per-rank frames -> packed tiles -> rank 0 frame, as bench.py does over RCCL
    tile, world = 64, 3
    scene = scenes.build_scene(L, d); cam = scenes.camera_of(d)
    dst = runtime.DeviceFramebuffer(d.width, d.height)
    n = runtime.packed_floats_per_rank(d.width, d.height, tile, world)
    for rank in range(world):
        fb = runtime.DeviceFramebuffer(d.width, d.height)
        runtime.render_device_sharded(L, cam, scene, fb, tile, rank, world)
        packed = torch.zeros(n, dtype=torch.float32, device="cuda")
        k = runtime.check(L.pack_tiles(fb.pixels.data_ptr(), fb.results.data_ptr(), d.width, d.height, 0, 0, d.width, d.height, tile, rank, world, packed.data_ptr(), None))
        assert k == len(runtime.shard_tiles(d.width, d.height, tile, rank, world))
        runtime.check(L.unpack_tiles(dst.pixels.data_ptr(), dst.results.data_ptr(), d.width, d.height, 0, 0, d.width, d.height, tile, rank, world, packed.data_ptr(), None))
    torch.cuda.synchronize()
    assert H.same_bits(dst.pixels_host(), whole["pixels"]) and np.array_equal(dst.results_host()["samples"], whole["samples"])
    L.scene_destroy(scene)


def test_errors_are_loud_on_the_gpu_box_too(H, L):
    d = scenes.cornell_box(16, 16, 1)
    scene = L.scene_create()
    obj = L.scene_add_object(scene, 2).contents
    scenes.fill_object(L, obj, scenes.ObjectDesc(d.objects[0].triangles[:2], d.objects[0].normals[:2], d.objects[0].texcoords[:2]))
    obj.material.bsdf.eval = C.cast(L.malloc, C.c_void_p)       # a foreign callback
    scenes.apply_options(L, scene, d)
    L.clear_error(); L.scene_commit(scene)
    assert "BSDF function pointers" in runtime.last_error()
    fb = api.Framebuffer(L, 16, 16); cam = scenes.camera_of(d)
    L.clear_error(); L.render(C.byref(cam), scene, C.byref(fb.fb), 0, 0, 16, 16)
    assert "no device replica" in runtime.last_error() and not fb.pixels.any()
    L.scene_destroy(scene)
    # Direct lighting without a light: the reference asserts (src/Terra.c:1617)
    dark = scenes.cornell_box(16, 16, 1, integrator=1); dark.objects[3].material.emissive = (0, 0, 0)
    s2 = scenes.build_scene(L, dark)
    L.clear_error(); L.render(C.byref(cam), s2, C.byref(fb.fb), 0, 0, 16, 16)
    assert "emissive" in runtime.last_error()
    L.scene_destroy(s2)


# ---------------------------------------------------------------------------
# full-size properties (BASELINE.json configs[1] geometry: 1920x1080 Cornell)
# ---------------------------------------------------------------------------

def test_full_hd_properties(H, L, orc_lib, devmath_mode):
    import torch
    spp = 16
    d = scenes.cornell_box(1920, 1080, spp)
    full = render_dev(L, d)
    s = full["stats"]
    assert s["samples"] == 1920 * 1080 * spp and s["pixels"] == 1920 * 1080
    assert s["rays"] >= s["samples"] and s["nodes"] >= s["rays"] and s["hits"] <= s["rays"] and s["rand_calls"] == 4 * (s["hits"]) - 0 * s["samples"]
    assert (full["samples"] == spp).all() and np.isfinite(full["pixels"]).all()
    # tile invariance: any tile rendered alone equals the same region of the full-frame render
    r = H.rng(17)
    scene = scenes.build_scene(L, d); cam = scenes.camera_of(d)
    fb = runtime.DeviceFramebuffer(1920, 1080)
    rects = [(int(r.randint(0, 1800)), int(r.randint(0, 1000)), int(r.randint(1, 120)), int(r.randint(1, 80))) for _ in range(6)]
    for rect in rects:
        runtime.render_device(L, cam, scene, fb, rect)
    torch.cuda.synchronize()
    px = fb.pixels_host(); sm = fb.results_host()["samples"]
    mask = np.zeros((1080, 1920), bool)
    for (x, y, w, h) in rects:
        # rectangles may overlap: overlapping pixels were rendered twice (8+8 samples, different streams) -> exclude them
        mask[y:y + h, x:x + w] = True
    once = mask & (sm == spp)
    assert once.sum() > 1000 and H.same_bits(px[once], full["pixels"][once])
    assert not px[~mask].any()
    # a crop against the oracle at the full-frame geometry (camera rays depend on the frame size)
    x0, y0, w, h = 900, 500, 48, 32
    want = H.Unit("orc").render_pixels(d, rect=(x0, y0, w, h), want_calls=False)
    assert H.same_bits(full["pixels"][y0:y0 + h, x0:x0 + w], want["pixels"][y0:y0 + h, x0:x0 + w])
    L.scene_destroy(scene)


# ---------------------------------------------------------------------------
# config 3 geometry: ~100k-triangle hall (LDS holds only a node prefix; triangles from global memory)
# ---------------------------------------------------------------------------

def test_hall_100k_goldens(H, L):
    g = G(H, "render_hall")
    d = scenes.sponza_hall(64, 36, 1)
    L.clear_error()
    scene = scenes.build_scene(L, d)
    assert runtime.last_error() == "", runtime.last_error()
    U = H.Unit("amd")
    nodes = U.bvh_nodes(scene)
    assert len(nodes) == int(g["bvh_nodes"]) and H.digest(nodes) == bytes(g["bvh_sha256"]).hex()
    o, dd = H.scene_rays(71, 512, box=((-9.5, 0.3, -4.5), (9.5, 7.5, 4.5)))
    found, prim, point = U.bvh_traverse(scene, o, dd)
    assert np.array_equal(found, g["trav_found"]) and np.array_equal(prim, g["trav_prim"]) and H.same_bits(point, g["trav_point"])
    L.scene_destroy(scene)
    for integ, (w, h, spp) in {0: (160, 90, 2), 1: (64, 36, 1)}.items():
        out = render_dev(L, scenes.sponza_hall(w, h, spp, integrator=integ), calls=True)
        assert H.same_bits(out["pixels"], g[f"i{integ}_pixels"]), integ
        assert np.array_equal(out["rand_calls"], g[f"i{integ}_calls"].astype(np.uint32)), integ
        host = render_host(L, scenes.sponza_hall(w, h, spp, integrator=integ))
        assert H.same_bits(host["pixels"], g[f"i{integ}_pixels"]), integ


def test_fast_tree_ray_by_ray_including_axis_parallel_rays(H, L):
    """The fast tree's traversal must return the reference tree's hit for ANY ray -- its boxes are stored as centre + half extent
    and tested with min/max that drop NaNs, so rays with zero direction components (infinite inverse direction) are the corner:
    without the clamp in traverse_fast_resume they stay correct but visit every box along their line (one such ray cost 40 ms)."""
    U = H.Unit("amd")
    r = np.random.default_rng(77)
    for name, d, box in (("hall", scenes.sponza_hall(64, 36, 1), ((-9.5, 0.3, -4.5), (9.5, 7.5, 4.5))), ("spheres", scenes.cornell_spheres(64, 36, 1), ((-0.9, 0.1, -0.9), (0.9, 1.9, 0.9)))):
        L.clear_error()
        scene = scenes.build_scene(L, d, tree_mode=1)
        assert runtime.last_error() == "", runtime.last_error()
        o, dd = H.scene_rays(5, 4096, box=box)
        o = np.ascontiguousarray(o, np.float32); dd = np.ascontiguousarray(dd, np.float32)
        # rays 1024.. : one or two direction components exactly zero; the last 256 start exactly on a vertex coordinate (origin in a box plane)
        for i in range(1024, 4096):
            k = r.integers(0, 3); dd[i, k] = 0.0
            if i >= 2560: dd[i, (k + 1 + r.integers(0, 2)) % 3] = 0.0
            if i >= 3584 and i % 2: dd[i] = -dd[i]
        verts = np.concatenate([np.asarray(ob.triangles, np.float32).reshape(-1, 3) for ob in d.objects])
        for i in range(3840, 4096): o[i, i % 3] = verts[r.integers(0, len(verts)), i % 3]
        nrm = np.linalg.norm(dd, axis=1, keepdims=True); dd = np.ascontiguousarray(dd / np.where(nrm == 0, 1, nrm), np.float32)
        f0, p0, pt0 = U.bvh_traverse(scene, o, dd)
        f1, p1, pt1, nodes = U.bvh_traverse_fast(scene, o, dd)
        assert np.array_equal(f0, f1), name
        hit = f0 != 0
        assert np.array_equal(p0[hit], p1[hit]) and H.same_bits(pt0[hit], pt1[hit]), name
        assert hit.sum() > 2000, name
        # the work of an axis-parallel ray stays in line with that of a general one (unclamped: ~20,000 nodes on the hall)
        assert nodes[1024:].max() <= 4 * max(200, int(nodes[:1024].max())), (name, int(nodes[:1024].max()), int(nodes[1024:].max()))
        L.scene_destroy(scene)


def test_hall_full_hd_crop_vs_oracle_and_tiles(H, L, orc_lib, devmath_mode):
    """BASELINE.json configs[2] geometry at 1080p (reduced spp): a crop against the oracle, tile invariance"""
    d = scenes.sponza_hall(1920, 1080, 2, integrator=2)
    full = render_dev(L, d)
    x0, y0, w, h = 1000, 600, 40, 24
    want = H.Unit("orc").render_pixels(d, rect=(x0, y0, w, h), want_calls=False)
    assert H.same_bits(full["pixels"][y0:y0 + h, x0:x0 + w], want["pixels"][y0:y0 + h, x0:x0 + w])
    part = render_dev(L, d, rect=(960, 512, 192, 128))
    assert H.same_bits(part["pixels"][512:640, 960:1152], full["pixels"][512:640, 960:1152])
    assert full["stats"]["samples"] == 1920 * 1080 * 2


# ---------------------------------------------------------------------------
# fast tree (terra_amd_set_tree_mode(1)): different tree and traversal order, same image
# ---------------------------------------------------------------------------

@pytest.mark.parametrize("name,integ", [("cornell", 0), ("cornell", 2), ("phong", 1), ("spheres", 2), ("hall", 0), ("hall", 1), ("soup", 2)])
def test_fast_tree_selects_the_same_hits(H, L, name, integ):
    if name == "cornell":
        d = scenes.cornell_box(96, 64, 4, integrator=integ)
    elif name == "phong":
        d = scenes.cornell_phong(96, 64, 3, integrator=integ)
    elif name == "spheres":
        d = scenes.cornell_spheres(96, 64, 3, integrator=integ)
    elif name == "hall":
        d = scenes.sponza_hall(160, 90, 2, integrator=integ)
    else:
        from test_oracle_vs_reference import soup_scene
        d = soup_scene(H, 2000, 21, integrator=integ); d.width, d.height, d.spp = 80, 48, 3
    a = render_dev(L, d, calls=True, passes=2)
    b = render_dev(L, d, calls=True, passes=2, tree_mode=1)
    assert np.array_equal(a["rand_calls"], b["rand_calls"])
    assert same(H, a["pixels"], b["pixels"]) and same(H, a["acc"], b["acc"])
    assert b["stats"]["nodes"] < a["stats"]["nodes"] or name in ("cornell", "phong")


# ---------------------------------------------------------------------------
# automatic traversal (the default): leaf-box cull on LDS-resident scenes, fast tree on the others, replica when the
# numeric containment check fails -- always the replica's image
# ---------------------------------------------------------------------------

@pytest.mark.parametrize("name,integ", [("cornell", 0), ("cornell", 1), ("cornell", 2), ("phong", 0), ("phong", 6), ("soup40", 2), ("soup9", 0)])
def test_leaf_box_cull_is_invisible_and_accounted(H, L, name, integ):
    from test_oracle_vs_reference import soup_scene
    if name == "cornell":
        d = scenes.cornell_box(112, 80, 4, integrator=integ)
    elif name == "phong":
        d = scenes.cornell_phong(96, 64, 3, integrator=integ)
    else:
        d = soup_scene(H, int(name[4:]), 33, integrator=integ); d.width, d.height, d.spp = 72, 48, 3
    a = render_dev(L, d, calls=True, passes=2, tree_mode=0)
    b = render_dev(L, d, calls=True, passes=2, tree_mode=2)
    assert same(H, a["pixels"], b["pixels"]) and same(H, a["acc"], b["acc"]) and np.array_equal(a["rand_calls"], b["rand_calls"])
    sa, sb = a["stats"], b["stats"]
    assert sa["tri_culled"] == 0 and sb["tri_culled"] > 0
    for k in ("rays", "hits", "rand_calls", "attr_fetches"):      # what the image determines: exact
        assert sa[k] == sb[k], k
    # the cull launches test boxes with t = fma(plane, inv, -(o * inv)) (conservative within the containment proof's budget, not the reference's arithmetic):
    # a grazing box may be decided differently, so node and leaf counts agree to a few parts per million, not to the last unit
    assert close_counts(sa["nodes"], sb["nodes"])
    assert close_counts(sb["tri_tests"] + sb["tri_culled"], sa["tri_tests"])            # every leaf the reference tests is either tested or culled
    assert sb["tri_tests"] < sa["tri_tests"]


def test_automatic_mode_falls_back_per_call_when_the_camera_is_far(H, L):
    """the containment argument covers ray origins inside the verified range only: a camera beyond it renders in replica mode"""
    d = scenes.cornell_box(64, 48, 2)
    d.camera_position = (0.0, 1.0, -40.0)
    a = render_dev(L, d, calls=True, tree_mode=0)
    b = render_dev(L, d, calls=True, tree_mode=2)
    assert same(H, a["pixels"], b["pixels"]) and b["stats"]["tri_culled"] == 0 and b["stats"]["tri_tests"] == a["stats"]["tri_tests"]
    hall = scenes.sponza_hall(64, 36, 1); hall.camera_position = (0.0, 4.0, -30.0)
    a = render_dev(L, hall, tree_mode=0); b = render_dev(L, hall, tree_mode=2)
    assert same(H, a["pixels"], b["pixels"]) and a["stats"]["nodes"] == b["stats"]["nodes"]       # the reference tree was traversed


def test_fast_tree_built_on_the_device(H, L):
    """terra_amd_set_tree_builder(1): LBVH built on the GPU from the soup in HBM -- another tree shape, the same image"""
    import torch
    from test_oracle_vs_reference import soup_scene
    g = G(H, "render_hall")
    cases = [scenes.sponza_hall(160, 90, 2, integrator=0), scenes.sponza_hall(64, 36, 1, integrator=1)]
    soup = soup_scene(H, 3000, 44, integrator=2); soup.width, soup.height, soup.spp = 80, 48, 3
    cases.append(soup)
    dup = soup_scene(H, 400, 45, integrator=0); dup.width, dup.height, dup.spp = 64, 40, 2
    for o in dup.objects:                                   # coincident triangles: equal Morton codes and equal depths (rank tie-breaks)
        o.triangles = np.concatenate([o.triangles, o.triangles]); o.normals = np.concatenate([o.normals, o.normals]); o.texcoords = np.concatenate([o.texcoords, o.texcoords])
    cases.append(dup)
    for k, d in enumerate(cases):
        outs = []
        for builder in (0, 1):
            L.clear_error()
            scene = scenes.build_scene(L, d, tree_mode=1, tree_builder=builder)
            assert runtime.last_error() == "", runtime.last_error()
            ti = runtime.TraversalInfo(); runtime.check(L.traversal_info(scene, C.byref(ti)))
            assert ti.fast_tree == 1 and ti.fast_tree_built_on_device == builder
            fb = runtime.DeviceFramebuffer(d.width, d.height); cam = scenes.camera_of(d)
            rc = torch.zeros(d.width * d.height, dtype=torch.int32, device="cuda")
            for _ in range(2):
                runtime.render_device(L, cam, scene, fb, None, rc)
            torch.cuda.synchronize()
            st = runtime.Stats(); runtime.check(L.get_stats(scene, C.byref(st)))
            outs.append((fb.pixels_host().copy(), fb.results_host()["acc"].copy(), rc.cpu().numpy().copy(), st.as_dict()))
            L.scene_destroy(scene)
        assert same(H, outs[0][0], outs[1][0]) and same(H, outs[0][1], outs[1][1]) and np.array_equal(outs[0][2], outs[1][2]), k
        assert outs[1][3]["nodes"] > 0 and outs[1][3]["rays"] == outs[0][3]["rays"] and outs[1][3]["hits"] == outs[0][3]["hits"]
    # ... and it is the reference's image: the hall golden (one pass)
    out = render_dev(L, scenes.sponza_hall(160, 90, 2, integrator=0), calls=True, tree_mode=1)
    scene = scenes.build_scene(L, scenes.sponza_hall(160, 90, 2, integrator=0), tree_mode=2, tree_builder=1)       # automatic mode: built on the device, read back, containment-checked
    ti = runtime.TraversalInfo(); runtime.check(L.traversal_info(scene, C.byref(ti)))
    assert ti.fast_tree_built_on_device == 1 and "device" in ti.note.decode()
    fb = runtime.DeviceFramebuffer(160, 90); runtime.render_device(L, scenes.camera_of(scenes.sponza_hall(160, 90, 2)), scene, fb); torch.cuda.synchronize()
    assert H.same_bits(fb.pixels_host(), g["i0_pixels"]) and H.same_bits(fb.pixels_host(), out["pixels"])
    L.scene_destroy(scene)


def test_reachability_mode_outside_the_coordinate_range(H, L, monkeypatch):
    """Coordinates beyond +-13 units: the automatic mode keeps the fast tree and accepts a hit only if the reference traversal would have reached it
    (reference_reaches: the inner ancestors' slab tests replayed). It must equal the replica bit for bit -- also when the reference's boxes are made to miss
    (test hook: the device copy of the reference tree shrunk), which float rounding alone does too rarely to test; the plain fast tree must then differ."""
    import torch
    from tools.scaled_hall import scaled

    def render(mode, integ, shrink=None, builder=None):
        d = scaled(scenes.sponza_hall(160, 90, 2, integrator=integ), 100.0)
        L.clear_error(); s = scenes.build_scene(L, d, tree_mode=mode, debug_shrink=shrink, tree_builder=builder)
        assert runtime.last_error() == "", runtime.last_error()
        ti = runtime.TraversalInfo(); runtime.check(L.traversal_info(s, C.byref(ti)))
        fb = runtime.DeviceFramebuffer(d.width, d.height)
        runtime.render_device(L, scenes.camera_of(d), s, fb); torch.cuda.synchronize()
        out = (fb.pixels_host().copy(), fb.results_host()["acc"].copy(), ti.fast_tree, ti.note.decode())
        L.scene_destroy(s)
        return out

    # the compiled reference's own image of the scaled hall (tests/golden/render_hall_x100.npz)
    g = G(H, "render_hall_x100")
    for integ, (w, h, spp) in {0: (160, 90, 2), 1: (64, 36, 1)}.items():
        out = render_dev(L, scaled(scenes.sponza_hall(w, h, spp, integrator=integ), 100.0), calls=True)
        assert H.same_bits(out["pixels"], g[f"i{integ}_pixels"]) and np.array_equal(out["rand_calls"], g[f"i{integ}_calls"].astype(np.uint32)), integ
    for shrink in (None, 3.0):
        for integ in (0, 1):
            ref = render(0, integ, shrink)
            for builder in (0, 1):          # host binned SAH, device LBVH (its boxes inflated on the device, the replay tables made from the read-back)
                auto = render(2, integ, shrink, builder)
                assert auto[2] == 1 and "reachability" in auto[3], auto[3]
                assert same(H, ref[0], auto[0]) and same(H, ref[1], auto[1]), (shrink, integ, builder)
        if shrink is not None:
            plain = render(1, 0, shrink); ref = render(0, 0, shrink)
            assert not same(H, plain[1], ref[1])          # the hook bites: without the replay the fast tree finds hits the (shrunk) reference misses
        # ... and ray by ray: the fast tree with the replay against the reference tree's own traversal
        d = scaled(scenes.sponza_hall(64, 36, 1), 100.0)
        s = scenes.build_scene(L, d, tree_mode=2, debug_shrink=shrink)
        o, dd = H.scene_rays(6, 4096, box=((-950.0, 30.0, -450.0), (950.0, 750.0, 450.0)))
        U = H.Unit("amd")
        f0, p0, pt0 = U.bvh_traverse(s, o, dd); f1, p1, pt1, _ = U.bvh_traverse_fast(s, o, dd)
        hit = f0 != 0
        assert np.array_equal(f0, f1) and np.array_equal(p0[hit], p1[hit]) and H.same_bits(pt0[hit], pt1[hit]) and hit.sum() > 1000, shrink
        L.scene_destroy(s)


def test_out_of_range_lds_resident_scene_keeps_the_leaf_box_cull(H, L, orc_lib, devmath_mode):
    """Coordinates beyond +-13 units, scene small enough to be staged in LDS (the Cornell box x 100): the reference tree is traversed decision by decision and
    only the LEAF boxes -- which the reference never tests -- are rebuilt around their triangles at the scene's rounding bound, so the cull skips nothing that
    could hit: equal to the replica and to the oracle bit for bit, with fewer triangle tests; also with the reference's inner boxes made to miss (test hook);
    a camera beyond TerraAmdTraversalInfo::camera_limit sends that call down the replica traversal and last_call says so."""
    import torch
    from tools.scaled_hall import scaled

    def render(mode, integ, shrink=None, cam_scale=1.0):
        d = scaled(scenes.cornell_box(96, 64, 6, integrator=integ), 100.0)
        d.camera_position = tuple(c * cam_scale for c in d.camera_position)
        L.clear_error(); s = scenes.build_scene(L, d, tree_mode=mode, debug_shrink=shrink)
        assert runtime.last_error() == "", runtime.last_error()
        fb = runtime.DeviceFramebuffer(d.width, d.height)
        rc = torch.zeros(d.width * d.height, dtype=torch.int32, device="cuda")
        runtime.render_device(L, scenes.camera_of(d), s, fb, None, rc); torch.cuda.synchronize()
        ti = runtime.TraversalInfo(); runtime.check(L.traversal_info(s, C.byref(ti)))
        st = runtime.Stats(); runtime.check(L.get_stats(s, C.byref(st)))
        out = dict(pixels=fb.pixels_host().copy(), acc=fb.results_host()["acc"].copy(), calls=rc.cpu().numpy().copy(), ti=(ti.fast_tree, ti.leaf_cull, ti.last_call, ti.note.decode(), ti.camera_limit), stats=st.as_dict(), d=d)
        L.scene_destroy(s)
        return out

    for integ in (0, 1, 2):
        for shrink in (None, 20.0):
            ref = render(0, integ, shrink); auto = render(2, integ, shrink)
            assert auto["ti"][:3] == (0, 1, 2) and "leaf boxes rebuilt" in auto["ti"][3] and ref["ti"][2] == 1, auto["ti"]
            assert same(H, ref["pixels"], auto["pixels"]) and same(H, ref["acc"], auto["acc"]) and np.array_equal(ref["calls"], auto["calls"]), (integ, shrink)
            assert auto["stats"]["tri_tests"] < ref["stats"]["tri_tests"] and close_counts(auto["stats"]["tri_tests"] + auto["stats"]["tri_culled"], ref["stats"]["tri_tests"])
            assert close_counts(auto["stats"]["nodes"], ref["stats"]["nodes"]) and auto["stats"]["hits"] == ref["stats"]["hits"]
        want = H.Unit("orc").render_pixels(ref["d"], threads=8)       # (unshrunk) the oracle's image of the scaled box
        plain = render(2, integ)
        assert same(H, plain["pixels"], want["pixels"]) and np.array_equal(plain["calls"].reshape(want["rand_calls"].shape).astype(np.uint64), want["rand_calls"].astype(np.uint64)), integ
    far = render(2, 0, cam_scale=12.0)           # 12 x (0, 100, -340): beyond 8 x the largest VERTEX coordinate (200)
    assert far["ti"][1] == 1 and far["ti"][2] == 1 and abs(far["ti"][4] - 8 * 200.0) < 1.0
    assert same(H, far["pixels"], render(0, 0, cam_scale=12.0)["pixels"]) and far["stats"]["tri_culled"] == 0


def test_host_fast_tree_build_does_not_depend_on_its_threads(H, L, monkeypatch):
    """the host builder hands subtrees to several threads (tree_build.cpp); the tree -- hence the work a render does -- must be the same whatever the schedule"""
    import torch
    d = scenes.sponza_hall(160, 90, 2, integrator=0)
    got = []
    for threads in (1, 3, 0, 0):           # terra_amd_set_build_threads (0 = every core the process may use)
        runtime.check(L.set_build_threads(threads))
        scene = scenes.build_scene(L, d, tree_mode=1, tree_builder=0)
        fb = runtime.DeviceFramebuffer(d.width, d.height)
        runtime.render_device(L, scenes.camera_of(d), scene, fb); torch.cuda.synchronize()
        st = runtime.Stats(); runtime.check(L.get_stats(scene, C.byref(st))); st = st.as_dict()
        got.append((fb.pixels_host().copy(), st["nodes"], st["tri_tests"], st["rays"]))
        L.scene_destroy(scene)
    for g in got[1:]:
        assert same(H, g[0], got[0][0]) and g[1:] == got[0][1:], (g[1:], got[0][1:])


def test_fast_tree_hall_goldens_and_work(H, L):
    """the fast tree reproduces the REFERENCE's image of the 97k-triangle hall with a fraction of the traversal work"""
    g = G(H, "render_hall")
    out = render_dev(L, scenes.sponza_hall(160, 90, 2, integrator=0), calls=True, tree_mode=1)
    assert H.same_bits(out["pixels"], g["i0_pixels"]) and np.array_equal(out["rand_calls"], g["i0_calls"].astype(np.uint32))
    ref = render_dev(L, scenes.sponza_hall(160, 90, 2, integrator=0))
    assert out["stats"]["nodes"] * 4 < ref["stats"]["nodes"]


# ---------------------------------------------------------------------------
# the work counters behind Mrays/s and the roofline's algorithmic bytes
# ---------------------------------------------------------------------------

@pytest.mark.parametrize("name,integ", [("cornell", 0), ("cornell", 1), ("phong", 2), ("hall", 0), ("hall", 1), ("hall", 2)])
def test_device_work_counters_equal_the_oracles(H, L, orc_lib, devmath_mode, name, integ):
    import ctypes as C
    d = {"cornell": scenes.cornell_box, "phong": scenes.cornell_phong}.get(name, None)
    d = d(80, 56, 3, integrator=integ) if d else scenes.sponza_hall(48, 27, 1, integrator=integ)
    got = render_dev(L, d)["stats"]

    class Ctr(C.Structure):
        _fields_ = [(n, C.c_uint64) for n in ("rays", "nodes", "box_tests", "tri_tests", "hits", "samples", "rand_calls", "attr_fetches")]
    orc_lib.fn("orc_counters_reset", None, [])()
    H.Unit("orc").render_pixels(d, want_calls=False)
    c = Ctr(); orc_lib.fn("orc_counters_get", None, [C.POINTER(Ctr)])(C.byref(c))
    want = {n: int(getattr(c, n)) for n, _ in Ctr._fields_}
    for k in want:
        assert got[k] == want[k], (k, got[k], want[k])
    assert got["pixels"] == d.width * d.height and got["launches"] == 1


# ---------------------------------------------------------------------------
# textured attributes on the device
# ---------------------------------------------------------------------------

def test_textured_attributes_match_the_reference(H, L):
    g = G(H, "render_textured")
    for integ in (0, 1, 2):
        d = scenes.cornell_textured(64, 48, 3, integrator=integ)
        L.clear_error()
        out = render_dev(L, d, passes=2, calls=True)
        assert runtime.last_error() == "", runtime.last_error()
        assert H.same_bits(out["pixels"], g[f"i{integ}_pixels"]), integ
        assert np.array_equal(out["rand_calls"], g[f"i{integ}_calls"].astype(np.uint32)), integ
        assert H.same_bits(render_dev(L, d, passes=2, tree_mode=1)["pixels"], g[f"i{integ}_pixels"]), integ


def test_mirror_addressing_and_texture_errors(H, L, orc_lib, devmath_mode):
    d = scenes.cornell_textured(64, 48, 2, integrator=1, mirror=True)      # the reference's mirror mode reads out of bounds: device vs oracle only
    assert H.same_bits(render_dev(L, d)["pixels"], H.Unit("orc").render_pixels(d, want_calls=False)["pixels"])
    # a lat-long lookup bound to a material attribute is rejected (it reads past the texcoord in the reference)
    import ctypes as C
    d2 = scenes.cornell_box(16, 16, 1)
    scene = L.scene_create()
    for od in d2.objects:
        scenes.fill_object(L, L.scene_add_object(scene, len(od.triangles)).contents, od)
    tex = api.TerraTexture(); data = np.zeros((2, 2, 3), np.float32)
    L.texture_init_hdr(C.byref(tex), 2, 2, 3, data.ctypes.data)
    obj = L.scene_add_object(scene, 1).contents
    scenes.fill_object(L, obj, scenes.ObjectDesc(d2.objects[0].triangles[:1], d2.objects[0].normals[:1], d2.objects[0].texcoords[:1]))
    a = api.TerraAttribute(); L.attribute_init_cubemap(C.byref(a), C.byref(tex))
    obj.material.attributes[0] = a
    scenes.apply_options(L, scene, d2)
    L.clear_error(); L.scene_commit(scene)
    assert "terra_texture_sample" in runtime.last_error()
    L.scene_destroy(scene)


# ---------------------------------------------------------------------------
# empty, degenerate and ragged inputs
# ---------------------------------------------------------------------------

def _tri_scene(tris, w=24, h=16, spp=2, integ=0, emissive=(3.0, 2.0, 1.0)):
    tris = np.asarray(tris, np.float32).reshape(-1, 3, 3)
    nrm = np.zeros_like(tris); nrm[..., 2] = -1
    objs = [scenes.ObjectDesc(tris, nrm, np.zeros((len(tris), 3, 2), np.float32), scenes.Material(albedo=(0.6, 0.5, 0.4), emissive=emissive))] if len(tris) else []
    return scenes.SceneDesc(objects=objs, width=w, height=h, spp=spp, bounces=3, integrator=integ, camera_position=(0.0, 0.0, -3.0), name="tris")


def test_empty_and_tiny_scenes(H, L, orc_lib, devmath_mode):
    big = [[-1, -1, 1], [1, -1, 1], [0, 1, 1]]
    cases = {
        "no objects": _tri_scene([]),
        "one triangle": _tri_scene([big]),
        "two triangles": _tri_scene([big, [[-1, -1, 2], [1, -1, 2], [0, 1, 2]]]),
        "degenerate triangle": _tri_scene([big, [[0, 0, 1], [0, 0, 1], [0, 0, 1]], [[-2, -2, 3], [2, -2, 3], [0, 2, 3]]]),
        "coincident triangles (depth ties)": _tri_scene([big, big, big], integ=1),
    }
    for name, d in cases.items():
        L.clear_error()
        got = render_dev(L, d, calls=True)
        assert runtime.last_error() == "", (name, runtime.last_error())
        want = H.Unit("orc").render_pixels(d)
        assert same(H, got["pixels"], want["pixels"]) and np.array_equal(got["rand_calls"], want["rand_calls"]), name
        assert (got["samples"] == d.spp).all(), name
    assert not render_dev(L, cases["no objects"])["pixels"].any()
    # an object with zero triangles next to a normal one
    d = _tri_scene([big])
    d.objects.append(scenes.ObjectDesc(np.zeros((0, 3, 3), np.float32), np.zeros((0, 3, 3), np.float32), np.zeros((0, 3, 2), np.float32)))
    assert same(H, render_dev(L, d)["pixels"], H.Unit("orc").render_pixels(d, want_calls=False)["pixels"])


def _light_tie_scene(integ, occluder_first, w=40, h=28, spp=3):
    """a light of COINCIDENT triangles (every triangle twice), a dark object that coincides with one of the light's triangles, a floor and a blocker: the light-sample
    rays of Direct / MIS meet several triangles at exactly the depth of the one they expect, and which of them the reference's traversal prefers decides the pixel"""
    def obj(tris, normal, mat):
        tris = np.asarray(tris, np.float32).reshape(-1, 3, 3); n = np.zeros_like(tris); n[...] = normal
        return scenes.ObjectDesc(tris, n, np.zeros((len(tris), 3, 2), np.float32), mat)
    l1, l2 = [[-1, 2, 0], [1, 2, 0], [1, 2, 2]], [[-1, 2, 0], [1, 2, 2], [-1, 2, 2]]
    light = obj([l1, l2, l1, l2], (0, -1, 0), scenes.Material(albedo=(0.5, 0.5, 0.5), emissive=(6.0, 5.0, 4.0)))
    dark = obj([l1], (0, -1, 0), scenes.Material(albedo=(0.3, 0.6, 0.3)))
    floor = obj([[[-2, 0, -1], [2, 0, -1], [2, 0, 3]], [[-2, 0, -1], [2, 0, 3], [-2, 0, 3]]], (0, 1, 0), scenes.Material(albedo=(0.7, 0.6, 0.5)))
    blocker = obj([[[-0.4, 1, 0.6], [0.5, 1, 0.6], [0.5, 1, 1.4]]], (0, 1, 0), scenes.Material(albedo=(0.6, 0.3, 0.3)))
    objs = [dark, light, floor, blocker] if occluder_first else [light, floor, blocker, dark]
    return scenes.SceneDesc(objects=objs, width=w, height=h, spp=spp, bounces=3, integrator=integ, camera_position=(0.0, 0.9, -3.0), name="light ties")


@pytest.mark.parametrize("integ", [1, 2])
def test_light_sample_rays_that_meet_triangles_at_the_depth_of_their_own(H, L, orc_lib, devmath_mode, integ):
    """the light-sample rays' shortcut (trace_device.h fast_expect / traverse_loops ANYHIT; kernels without counters) against the oracle where it is easiest to get
    wrong: depth ties with the expected triangle, resolved by the reference's visit order"""
    for occluder_first in (False, True):
        d = _light_tie_scene(integ, occluder_first)
        want = H.Unit("orc").render_pixels(d, want_calls=False)
        assert want["pixels"].any()
        for mode in (0, 1, 2):              # reference tree from LDS (replica), fast tree, automatic (LDS-resident: leaf-box cull)
            for counters in (True, False):
                got = render_dev(L, d, tree_mode=mode, counters=counters)
                assert same(H, got["pixels"], want["pixels"]) and same(H, got["acc"], want["acc"]), (occluder_first, mode, counters)


@pytest.mark.parametrize("counters", [False])
def test_the_kernels_without_counters_render_the_same_frames(H, L, orc_lib, devmath_mode, counters):
    """most tests here read the work counters and so run the counting instantiations; this one runs what a client gets: every integrator family, LDS-resident and
    global-memory scenes, the three tree modes -- against the oracle"""
    from test_oracle_vs_reference import soup_scene
    cases = [scenes.cornell_box(48, 32, 3, integrator=i) for i in (0, 1, 2)] + [scenes.cornell_phong(40, 28, 2, integrator=1), scenes.cornell_spheres(40, 28, 2, integrator=2)]
    for i in (0, 1, 2):
        big = soup_scene(H, 1500, 90 + i, integrator=i); big.width, big.height, big.spp = 48, 32, 2
        cases.append(big)
    cases.append(scenes.sponza_hall(64, 36, 1, integrator=1))
    for k, d in enumerate(cases):
        want = H.Unit("orc").render_pixels(d, want_calls=False)
        for mode in (0, 1, 2):
            got = render_dev(L, d, tree_mode=mode, counters=counters)
            assert got["stats"]["rays"] == 0                       # (the device counters really were off)
            assert same(H, got["pixels"], want["pixels"]) and same(H, got["acc"], want["acc"]), (k, d.name, mode)


def test_ragged_frames_and_tiles(H, L, orc_lib, devmath_mode):
    for (w, h) in ((1, 1), (1, 37), (63, 1), (17, 9), (65, 65)):
        d = scenes.cornell_box(w, h, 2, integrator=1)
        want = H.Unit("orc").render_pixels(d, want_calls=False)
        assert same(H, render_dev(L, d)["pixels"], want["pixels"]), (w, h)
        assert same(H, render_host(L, d)["pixels"], want["pixels"]), (w, h)
    d = scenes.cornell_box(70, 50, 2)
    whole = render_dev(L, d)
    rects = [(0, 0, 1, 1), (69, 49, 1, 1), (3, 5, 64, 1), (5, 0, 1, 50), (7, 9, 33, 17)]
    for r in rects:
        part = render_dev(L, d, rect=r)
        x, y, w, h = r
        assert H.same_bits(part["pixels"][y:y + h, x:x + w], whole["pixels"][y:y + h, x:x + w]), r
        mask = np.ones((50, 70), bool); mask[y:y + h, x:x + w] = False
        assert not part["pixels"][mask].any() and not part["samples"][mask].any(), r
    # rectangles outside the frame are refused
    import ctypes as C
    scene = scenes.build_scene(L, d); cam = scenes.camera_of(d); fb = runtime.DeviceFramebuffer(70, 50)
    assert L.render_device(C.byref(cam), scene, fb.pixels.data_ptr(), fb.results.data_ptr(), 70, 50, 60, 0, 11, 5, None, None) < 0
    assert L.render_device(C.byref(cam), scene, fb.pixels.data_ptr(), fb.results.data_ptr(), 70, 50, 0, 0, 0, 5, None, None) < 0
    # the job table of LDS-resident launches packs a pixel into 32 bits: a frame with a side above 65,535 is refused with a message, a side of 65,535 renders
    wide = runtime.DeviceFramebuffer(65536, 1)
    L.clear_error()
    assert L.render_device(C.byref(cam), scene, wide.pixels.data_ptr(), wide.results.data_ptr(), 65536, 1, 65000, 0, 536, 1, None, None) < 0 and "65,535" in runtime.last_error()
    L.clear_error()
    ok = runtime.DeviceFramebuffer(65535, 1)
    assert L.render_device(C.byref(cam), scene, ok.pixels.data_ptr(), ok.results.data_ptr(), 65535, 1, 65000, 0, 535, 1, None, None) == 0
    import torch; torch.cuda.synchronize()
    assert (ok.results_host()["samples"][0, 65000:] == 2).all()
    L.scene_destroy(scene)


def test_scene_lifecycle_recommit_clear_and_many_objects(H, L, orc_lib, devmath_mode):
    import ctypes as C
    d = scenes.cornell_box(40, 30, 2)
    scene = scenes.build_scene(L, d); cam = scenes.camera_of(d)
    fb = runtime.DeviceFramebuffer(40, 30)
    runtime.render_device(L, cam, scene, fb)
    # options are double buffered: editing them has no effect until the next commit (reference src/Terra.c:182,251-255)
    L.scene_get_options(scene).contents.samples_per_pixel = 5
    runtime.render_device(L, cam, scene, fb)
    assert (fb.results_host()["samples"] == 4).all()
    L.scene_commit(scene)
    runtime.render_device(L, cam, scene, fb)
    assert (fb.results_host()["samples"] == 9).all()
    # clear + refill + commit builds a new replica
    L.scene_clear(scene)
    assert L.scene_count_objects(scene) == 0
    d2 = scenes.cornell_phong(40, 30, 2, integrator=1)
    for od in d2.objects:
        scenes.fill_object(L, L.scene_add_object(scene, len(od.triangles)).contents, od)
    scenes.apply_options(L, scene, d2)
    L.clear_error(); L.scene_commit(scene)
    assert runtime.last_error() == ""
    fb.clear(); runtime.render_device(L, cam, scene, fb)
    assert same(H, fb.pixels_host(), H.Unit("orc").render_pixels(d2, want_calls=False)["pixels"])
    L.scene_destroy(scene)
    # 256 objects is the limit of the 8-bit object index (reference include/Terra.h:195-198); 257 is refused
    tri = d.objects[0]
    for n, ok in ((256, True), (257, False)):
        s2 = L.scene_create()
        for k in range(n):
            od = scenes.ObjectDesc(tri.triangles[:1] + np.float32(0.001 * k), tri.normals[:1], tri.texcoords[:1], scenes.Material(emissive=(1, 1, 1) if k == 0 else (0, 0, 0)))
            scenes.fill_object(L, L.scene_add_object(s2, 1).contents, od)
        scenes.apply_options(L, s2, d)
        L.clear_error(); L.scene_commit(s2)
        assert (runtime.last_error() == "") == ok, (n, runtime.last_error())
        L.scene_destroy(s2)


def test_automatic_tree_mode_picks_by_scene_size(H, L):
    """mode 2: the reference tree for LDS-resident scenes, the fast tree otherwise; same image either way"""
    for mk, expect_fast in ((lambda: scenes.cornell_box(64, 48, 2), False), (lambda: scenes.cornell_spheres(64, 48, 2), True)):
        ref, fast, auto = (render_dev(L, mk(), tree_mode=m) for m in (0, 1, 2))
        assert same(H, auto["pixels"], ref["pixels"]) and same(H, fast["pixels"], ref["pixels"])
        assert auto["stats"]["nodes"] == (fast if expect_fast else ref)["stats"]["nodes"]
    scene = L.scene_create()
    assert L.set_tree_mode(scene, 3) < 0
    L.clear_error(); L.scene_destroy(scene)


# ---------------------------------------------------------------------------
# decoupled loop (scenes read from global memory, Simple/debug integrators): lanes of a wave are on
# different rays; per pixel nothing may change
# ---------------------------------------------------------------------------

@pytest.mark.parametrize("name,integ", [("spheres", 0), ("spheres", 4), ("spheres", 5), ("soup", 0), ("soup", 3), ("hall", 4), ("soup", 1), ("hall", 1)])
def test_decoupled_loop_matches_the_oracle(H, L, orc_lib, devmath_mode, name, integ):
    if name == "spheres":       # GGX + glass + diffuse: the generic kinds, 3,980 triangles, ragged frame, environment term on
        d = scenes.cornell_spheres(104, 72, 3, integrator=integ, environment=(0.3, 0.4, 0.9), environment_lighting=True)
    elif name == "hall":
        d = scenes.sponza_hall(72, 40, 2, integrator=integ)
    else:
        from test_oracle_vs_reference import soup_scene
        d = soup_scene(H, 1500, 33, integrator=integ); d.width, d.height, d.spp = 88, 56, 3
    want = H.Unit("orc").render_pixels(d, passes=2)
    got = render_dev(L, d, calls=True, passes=2)
    assert same(H, got["pixels"], want["pixels"]) and same(H, got["acc"], want["acc"])
    scene = scenes.build_scene(L, d)
    info = runtime.SceneInfo(); runtime.check(L.scene_info(scene, C.byref(info)))
    L.scene_destroy(scene)
    assert info.triangles > 300, "must be a scene the LDS plan does not stage whole"


# ---------------------------------------------------------------------------
# a traversal stack deeper than 64 KB of LDS per block (render_kernels.hip launch_instance opts in; terra_plan_lds clamps the leaf list). The reference's own builder
# only produces such stacks on inputs of pathological size, so the launches are driven through the test hook terra_amd_debug_pad_stack on an ordinary scene.
# ---------------------------------------------------------------------------

def _padded_render(L, d, mode, pad, fast_lds=0):
    scene = scenes.build_scene(L, d, tree_mode=mode)
    runtime.check(L.debug_pad_stack(scene, pad))
    runtime.check(L.debug_fast_stack_lds(scene, fast_lds))
    fb = runtime.DeviceFramebuffer(d.width, d.height)
    L.clear_error()
    rc = L.render_device(C.byref(scenes.camera_of(d)), scene, fb.pixels.data_ptr(), fb.results.data_ptr(), d.width, d.height, 0, 0, d.width, d.height, None, None)
    import torch
    torch.cuda.synchronize()
    out = dict(rc=rc, err=runtime.last_error(), pixels=fb.pixels_host().copy(), acc=fb.results_host()["acc"].copy())
    L.clear_error(); L.scene_destroy(scene)
    return out


@pytest.mark.parametrize("integ", [0, 1])
def test_a_stack_deeper_than_64_kb_of_lds_renders(H, L, orc_lib, devmath_mode, integ):
    from test_oracle_vs_reference import soup_scene
    d = soup_scene(H, 1500, 77, integrator=integ); d.width, d.height, d.spp = 64, 40, 2          # not LDS-resident: reference tree from global memory / fast tree
    want = H.Unit("orc").render_pixels(d)
    for mode in (0, 2):
        for pad in (0, 60, 110):             # 60: the block asks for more than 64 KB (replica: stack + a 4-entry leaf list + parked rows); 110: ~130 KB, one block per CU
            got = _padded_render(L, d, mode, pad)
            assert got["rc"] == 0 and got["err"] == "", (mode, pad, got["err"])
            assert same(H, got["pixels"], want["pixels"]) and same(H, got["acc"], want["acc"]), (mode, pad)
    # the fast tree's stack keeps its first entries in LDS and the rest in HBM: with 1, 2 or 3 entries in LDS nearly every push of this scene goes to HBM
    for fast_lds in (1, 2, 3):
        got = _padded_render(L, d, 1, 0, fast_lds=fast_lds)
        assert got["rc"] == 0 and got["err"] == "", (fast_lds, got["err"])
        assert same(H, got["pixels"], want["pixels"]) and same(H, got["acc"], want["acc"]), fast_lds
    got = _padded_render(L, d, 0, 400)         # beyond what a block can hold: refused with a message, nothing launched, nothing rendered
    assert got["rc"] < 0 and "LDS per block" in got["err"] and not got["acc"].any()


def test_note_and_last_call_agree(H, L):
    """whatever the commit decided and the note says, terra_amd_traversal_info's flags and last_call name the traversal a render actually runs"""
    from test_oracle_vs_reference import soup_scene
    big = soup_scene(H, 1500, 78); big.width, big.height, big.spp = 32, 24, 1
    for d, mode in ((big, 2), (big, 0), (big, 1), (scenes.cornell_box(32, 24, 1), 2), (scenes.cornell_box(32, 24, 1), 0), (scenes.cornell_spheres(32, 24, 1), 2)):
        scene = scenes.build_scene(L, d, tree_mode=mode)
        fb = runtime.DeviceFramebuffer(d.width, d.height)
        runtime.render_device(L, scenes.camera_of(d), scene, fb)
        ti = runtime.TraversalInfo(); runtime.check(L.traversal_info(scene, C.byref(ti)))
        note = ti.note.decode(); name = runtime.CALL_TRAVERSAL[ti.last_call]
        if ti.fast_tree:
            assert "fast tree" in note and name.startswith("fast tree"), (note, name)
        elif ti.leaf_cull:
            assert "leaf-box cull" in note and name == "reference tree + leaf-box cull", (note, name)
        else:
            assert "replica" in note and name == "reference tree, replica traversal", (note, name)
        L.scene_destroy(scene)
