"""BASELINE.json's configurations at their FULL sizes through the C-ABI, in the traversal mode bench.py times (the library
default, terra_amd_set_tree_mode 2 = automatic; the replica and forced-fast modes ride along as extra cases), each checked the
two ways the domain allows:
a crop of the finished frame against the oracle (bit-exact: sums, tonemapped pixels, per-pixel stream-B draw totals
where the launch reports them), and size-independent properties of the whole frame (sample counts, determinism of the
shard -> pack -> unpack path, counter identities).

  configs[1]  Cornell box 1920x1080, 512 spp, 8 bounces -- the exact bench.py launch: its sample split (bench.DEFAULT_SPLIT = 0 = the library's automatic choice, 8 for this
              frame), i.e. the framebuffer of 8 successive calls of 64 spp (reference src/Terra.c:551-572: a call sums its samples from
              zero, adds them to the running sum and re-tonemaps)
  configs[2]  ~100k-triangle hall 1920x1080, 256 spp: automatic (what bench.py times), replica and forced fast tree;
              plus the hall x 100 (outside the containment range: reachability mode) at bench.py's 1080p / 64 spp
  configs[3]  Cornell + glass + GGX spheres 1920x1080, 1024 spp (PARITY UNPINNED: these two presets have no runnable
              reference form, src/TerraPresets.c:298-465 is #if 0; the oracle is this repo's definition)
  configs[4]  the hall at 3840x2160, 4096 spp, rendered as the 8 tile shards of the 8-GPU job (one after the other on
              this one GPU, each into its own frame), packed, unpacked into rank 0's frame
"""
import ctypes as C
import os

import numpy as np
import pytest

from terra_amd import api, runtime, scenes

pytestmark = pytest.mark.gpu

import bench  # noqa: E402  (nothing GPU-related at module level)

TILE = 64       # bench.py's tile size
SPLIT = 32           # the sample split bench.py gives its hall / sphere-scene workloads (the framebuffer of 32 successive calls of spp/32 samples)
HEADLINE_SPLIT = 8   # ... and what its default (0: the library's automatic split) resolves to for the headline frame on one GPU (checked below)
THREADS = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))


@pytest.fixture(scope="module")
def L(amd_lib):
    lib = runtime.load()
    assert lib.device_count() > 0, "gpu tests need a visible MI355X: " + runtime.last_error()
    return lib


def crop(a, rect):
    x, y, w, h = rect
    return a[y:y + h, x:x + w]


def assert_crop_equals_oracle(H, got, want, rect, calls=True):
    for k in ("acc", "pixels"):
        assert H.same_bits(crop(got[k], rect), crop(want[k], rect)), k
    assert np.array_equal(crop(got["samples"], rect), crop(want["samples"], rect))
    if calls:
        assert np.array_equal(crop(got["rand_calls"], rect).astype(np.uint64), crop(want["rand_calls"], rect).astype(np.uint64))


def traversal_name(ti):
    """the traversal as bench.py's config.traversal names it (bench.result_block)"""
    note = ti.note.decode()
    if ti.fast_tree:
        return "fast tree + reachability replay" if "reachability" in note else "fast tree"
    if ti.leaf_cull:
        return "reference tree + leaf-box cull + reachability replay" if "reachability" in note else "reference tree + leaf-box cull"
    return "reference tree, replica traversal"


def device_frame(L, d, split=1, tree_mode=None, calls=True, shard=None, counters=True):
    """one terra_amd_render_device call over the whole frame; returns host copies + the launch's work counters.
    tree_mode None = the library default (2, automatic): what bench.py times with --tree auto"""
    import torch
    L.clear_error()
    scene = scenes.build_scene(L, d, tree_mode=tree_mode, counters=counters)
    assert runtime.last_error() == "", runtime.last_error()
    runtime.check(L.set_sample_split(scene, split), "terra_amd_set_sample_split")
    ti = runtime.TraversalInfo(); runtime.check(L.traversal_info(scene, C.byref(ti)))
    fb = runtime.DeviceFramebuffer(d.width, d.height); cam = scenes.camera_of(d)
    rc = torch.zeros(d.width * d.height, dtype=torch.int32, device="cuda") if calls else None
    runtime.render_device(L, cam, scene, fb, None, rc)
    torch.cuda.synchronize()
    res = fb.results_host()
    out = dict(pixels=fb.pixels_host().copy(), acc=res["acc"].copy(), samples=res["samples"].copy())
    if calls:
        out["rand_calls"] = rc.cpu().numpy().astype(np.uint32).reshape(d.height, d.width)
    st = runtime.Stats(); runtime.check(L.get_stats(scene, C.byref(st))); out["stats"] = st.as_dict()
    info = runtime.SceneInfo(); runtime.check(L.scene_info(scene, C.byref(info))); out["triangles"] = info.triangles
    out["tree_mode"] = ti.tree_mode; out["traversal"] = traversal_name(ti)
    if hasattr(ti, "last_call"):
        runtime.check(L.traversal_info(scene, C.byref(ti))); out["last_call"] = ti.last_call
    L.scene_destroy(scene)
    return out


def test_config2_headline_launch_cornell_1080p_512spp_bench_split(H, L, orc_lib, devmath_mode):
    """the launch bench.py times: 1920x1080, 512 spp, its sample split (bench.DEFAULT_SPLIT = 0 -> 8: == 8 reference calls of 64 spp), library defaults"""
    d = scenes.cornell_box(1920, 1080, 512, bounces=8)
    assert bench.DEFAULT_SPLIT == 0 and L.auto_sample_split(1920, 1080, TILE, 1, 512, 1) == HEADLINE_SPLIT          # what bench.py's measure() resolves its default to
    assert [L.auto_sample_split(1920, 1080, TILE, n, 512, 1) for n in (2, 4, 8)] == [16, 32, 32] and L.auto_sample_split(1920, 1080, TILE, 1, 256, 0) == 16
    got = device_frame(L, d, split=HEADLINE_SPLIT, calls=False, counters=False)      # exactly bench.py's timed launch: library defaults -- automatic traversal, no work counters -- and its sample split
    assert got["tree_mode"] == 2 and got["traversal"] == "reference tree + leaf-box cull"      # what BENCH's config.traversal names
    assert (got["samples"] == 512).all() and np.isfinite(got["pixels"]).all()
    assert got["stats"]["samples"] == 1920 * 1080 * 512 and got["stats"]["pixels"] == 1920 * 1080 and got["stats"]["rays"] == 0      # (host-kept totals; the device counters are off)
    lean = device_frame(L, d, split=HEADLINE_SPLIT, calls=False)           # bench.py's extra counting launch: the same frame, counters on
    assert H.same_bits(lean["acc"], got["acc"]) and H.same_bits(lean["pixels"], got["pixels"])
    s = lean["stats"]
    assert s["samples"] == 1920 * 1080 * 512 and s["pixels"] == 1920 * 1080 and s["rand_calls"] == 4 * s["hits"] and s["rays"] > s["samples"]
    counted = device_frame(L, d, split=HEADLINE_SPLIT, calls=True)         # ... and with per-pixel stream-B draw counts
    assert H.same_bits(counted["acc"], got["acc"]) and H.same_bits(counted["pixels"], got["pixels"])
    assert int(counted["rand_calls"].astype(np.uint64).sum()) == s["rand_calls"]
    got["rand_calls"] = counted["rand_calls"]
    # two crops: inside the box (back wall), and the box's left border (background | red wall)
    dch = scenes.cornell_box(1920, 1080, 512 // HEADLINE_SPLIT, bounces=8)
    for rect in ((936, 300, 48, 32), (400, 520, 48, 32)):
        want = H.Unit("orc").render_pixels(dch, passes=HEADLINE_SPLIT, rect=rect, threads=THREADS, sum_calls=True)
        assert (crop(want["samples"], rect) == 512).all()
        assert_crop_equals_oracle(H, got, want, rect)
    # the same frame without the split is a DIFFERENT (equally valid) frame: one call of 512 spp; its crop is the oracle's too
    one = device_frame(L, d, split=1)
    rect = (936, 300, 48, 32)
    want1 = H.Unit("orc").render_pixels(d, passes=1, rect=rect, threads=THREADS)
    assert_crop_equals_oracle(H, one, want1, rect)
    assert not H.same_bits(crop(one["acc"], rect), crop(got["acc"], rect))
    # the replica traversal (mode 0) renders the same split-8 frame, and is the mode whose work counters are the reference's
    rep = device_frame(L, d, split=HEADLINE_SPLIT, tree_mode=0, calls=False)
    assert rep["traversal"] == "reference tree, replica traversal"
    assert H.same_bits(rep["acc"], got["acc"]) and H.same_bits(rep["pixels"], got["pixels"])
    # (the cull launch's fused box test may decide a grazing box differently from the replica: node counts agree to parts per million; hits are the image's)
    assert abs(rep["stats"]["nodes"] - s["nodes"]) <= 2e-5 * s["nodes"] and rep["stats"]["hits"] == s["hits"] and rep["stats"]["tri_tests"] > s["tri_tests"]


_hall_want = {}


@pytest.mark.parametrize("tree_mode", [None, 0, 1])
def test_config3_hall_1080p_256spp(H, L, orc_lib, devmath_mode, tree_mode):
    """None = the library default (automatic) with bench.py's sample split: the launch its `hall_1080p_256spp` workload times
    (= the framebuffer of 16 successive calls of 16 spp)"""
    d = scenes.sponza_hall(1920, 1080, 256, bounces=8)
    got = device_frame(L, d, split=SPLIT, tree_mode=tree_mode)
    assert got["traversal"] == {None: "fast tree", 0: "reference tree, replica traversal", 1: "fast tree"}[tree_mode]
    assert got["tree_mode"] == (2 if tree_mode is None else tree_mode)
    assert 90_000 <= got["triangles"] <= 110_000
    assert (got["samples"] == 256).all() and np.isfinite(got["pixels"]).all()
    assert got["stats"]["samples"] == 1920 * 1080 * 256 and got["stats"]["rand_calls"] == 4 * got["stats"]["hits"]
    rect = (1000, 600, 48, 32)
    if "w" not in _hall_want:                                   # one oracle run serves all tree modes (0.4 M samples of 474 nodes/ray each)
        _hall_want["w"] = H.Unit("orc").render_pixels(scenes.sponza_hall(1920, 1080, 256 // SPLIT, bounces=8), passes=SPLIT, rect=rect, threads=THREADS, sum_calls=True)
    assert_crop_equals_oracle(H, got, _hall_want["w"], rect)


def test_hall_x100_1080p_64spp_reachability_mode(H, L, orc_lib, devmath_mode):
    """bench.py's `hall_x100_1080p_64spp` workload at its full size: every coordinate (scene and camera) x 100, i.e. outside the
    +-13-unit range of the containment proof -- the automatic mode keeps the fast tree and replays the reference's reachability"""
    d = bench.workload("hall_x100_1080p_64spp")
    assert (d.width, d.height, d.spp) == (1920, 1080, 64)
    assert [w for w in bench.EXTRA_WORKLOADS if w[0] == "hall_x100_1080p_64spp"][0][3] == 4 and [w for w in bench.EXTRA_WORKLOADS if w[0] == "hall_1080p_256spp"][0][3] == SPLIT
    got = device_frame(L, d, split=4)                           # bench.py's launch: sample split 4 = the framebuffer of 4 successive calls of 16 spp
    assert got["tree_mode"] == 2 and got["traversal"] == "fast tree + reachability replay"
    assert (got["samples"] == 64).all() and np.isfinite(got["pixels"]).all()
    assert got["stats"]["samples"] == 1920 * 1080 * 64 and got["stats"]["rand_calls"] == 4 * got["stats"]["hits"]
    d16 = bench.workload("hall_x100_1080p_64spp", 16)
    for rect in ((1000, 600, 48, 32), (300, 820, 32, 16)):
        want = H.Unit("orc").render_pixels(d16, passes=4, rect=rect, threads=THREADS, sum_calls=True)
        assert_crop_equals_oracle(H, got, want, rect)


def test_config4_spheres_1080p_1024spp_unpinned(H, L, orc_lib, devmath_mode):
    """GGX + glass: device == oracle bit for bit, but the oracle's definition of these presets is this repo's (parity unpinned)"""
    d = scenes.cornell_spheres(1920, 1080, 1024, bounces=8)
    got = device_frame(L, d, split=SPLIT)                      # library default + bench.py's sample split: its `spheres_1080p_1024spp` launch
    assert got["tree_mode"] == 2 and got["traversal"] == "fast tree"
    assert (got["samples"] == 1024).all()
    assert got["stats"]["samples"] == 1920 * 1080 * 1024
    dch = scenes.cornell_spheres(1920, 1080, 1024 // SPLIT, bounces=8)
    for rect in ((1100, 700, 48, 32), (960, 760, 48, 32)):      # inside the glass sphere; metal sphere | gap | glass sphere silhouettes
        want = H.Unit("orc").render_pixels(dch, passes=SPLIT, rect=rect, threads=THREADS, sum_calls=True)
        assert_crop_equals_oracle(H, got, want, rect)
    # the reference-tree kernel gives the same frame (thinner: 64 spp, the same streams as the first of the 8 passes would not
    # be comparable, so both trees are rendered at 64 spp)
    a = device_frame(L, scenes.cornell_spheres(1920, 1080, 64), tree_mode=0)
    for mode in (1, None):
        b = device_frame(L, scenes.cornell_spheres(1920, 1080, 64), tree_mode=mode)
        assert H.same_bits(a["acc"], b["acc"]) and np.array_equal(a["rand_calls"], b["rand_calls"])


def _sharded_frame(L, d, world, split, tree_mode=None):
    """what the N-rank job does (bench.py / runtime.gather_frame), run rank after rank on one GPU: every rank renders its
    tiles into its OWN frame, packs them; rank 0 unpacks the peers' buffers into its frame"""
    import torch
    scene = scenes.build_scene(L, d, tree_mode=tree_mode)
    runtime.check(L.set_sample_split(scene, split))
    ti = runtime.TraversalInfo(); runtime.check(L.traversal_info(scene, C.byref(ti)))
    cam = scenes.camera_of(d)
    n = runtime.packed_floats_per_rank(d.width, d.height, TILE, world)
    dst = None
    own = 0
    for rank in range(world):
        fb = runtime.DeviceFramebuffer(d.width, d.height)
        runtime.render_device_sharded(L, cam, scene, fb, TILE, rank, world)
        if rank == 0:
            dst = fb
            own = int((fb.results_host()["samples"] > 0).sum())
            continue
        packed = torch.zeros(n, dtype=torch.float32, device="cuda")
        k = runtime.check(L.pack_tiles(fb.pixels.data_ptr(), fb.results.data_ptr(), d.width, d.height, 0, 0, d.width, d.height, TILE, rank, world, packed.data_ptr(), None))
        assert k == len(runtime.shard_tiles(d.width, d.height, TILE, rank, world))
        runtime.check(L.unpack_tiles(dst.pixels.data_ptr(), dst.results.data_ptr(), d.width, d.height, 0, 0, d.width, d.height, TILE, rank, world, packed.data_ptr(), None))
        torch.cuda.synchronize()
        del fb, packed
    torch.cuda.synchronize()
    res = dst.results_host()
    out = dict(pixels=dst.pixels_host().copy(), acc=res["acc"].copy(), samples=res["samples"].copy(), rank0_pixels=own, traversal=traversal_name(ti))
    L.scene_destroy(scene)
    return out


def test_config5_hall_2160p_4096spp_eight_shards(H, L, orc_lib, devmath_mode):
    """3840x2160, 4096 spp, 8-way tile shard (t % 8 == rank, 64-px tiles), gather emulated on one GPU.
    Full spp in the library-default automatic mode = the fast tree (34 G samples); the replica traversal at the same size is covered at 8 spp."""
    W, Ht, world = 3840, 2160, 8
    d = scenes.sponza_hall(W, Ht, 4096, bounces=8)
    got = _sharded_frame(L, d, world, split=SPLIT)             # library default (automatic): the fast tree; bench.py's sample split
    assert got["traversal"] == "fast tree"
    assert (got["samples"] == 4096).all() and np.isfinite(got["pixels"]).all()
    tiles = -(-W // TILE) * -(-Ht // TILE)
    assert abs(got["rank0_pixels"] - W * Ht / world) <= 2 * TILE * TILE * (tiles % world + 1)      # the shard rule deals tiles evenly
    dch = scenes.sponza_hall(W, Ht, 4096 // SPLIT, bounces=8)
    # (the oracle walks the reference tree: 474 nodes per ray, so the crops are small)
    for rect in ((2000, 1200, 16, 8),          # inside tile (31, 18)
                 (2040, 1212, 16, 8)):         # straddles the tile borders x = 2048 and y = 1216: four tiles of four different ranks
        want = H.Unit("orc").render_pixels(dch, passes=SPLIT, rect=rect, threads=THREADS, want_calls=False)
        assert_crop_equals_oracle(H, got, want, rect, calls=False)
    # sharded == unsharded, every traversal mode, whole 4K frame (size-independent: 8 spp)
    for tree_mode in (None, 0, 1):
        d8 = scenes.sponza_hall(W, Ht, 8, bounces=8)
        a = _sharded_frame(L, d8, world, split=1, tree_mode=tree_mode)
        b = device_frame(L, d8, split=1, tree_mode=tree_mode, calls=False)
        assert H.same_bits(a["acc"], b["acc"]) and H.same_bits(a["pixels"], b["pixels"]) and np.array_equal(a["samples"], b["samples"])
