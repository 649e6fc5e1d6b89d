"""Generates the golden vectors in this directory from the COMPILED REFERENCE
(oracle/_ref/libterra_ref.so, built by `make -C oracle ref` from the unmodified
sources under /root/reference with entropy pinned -- see oracle/ref_wrapper.c).

Run in the build container only:   python tests/golden/generate.py

What is stored is data: inputs and the reference's outputs (SURVEY.md section 4's
pin list). No reference source travels. The reference has no tests, golden
images or known-answer vectors of its own, so these are the
pins of the oracle: tests/test_oracle_golden.py replays them bit-for-bit.
"""
from __future__ import annotations

import ctypes as C
import json
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent))
sys.path.insert(0, str(HERE.parent.parent))
import harness as H  # noqa: E402
from terra_amd import api, scenes  # noqa: E402


def save(name, **arrays):
    np.savez_compressed(HERE / f"{name}.npz", **arrays)
    return {k: {"shape": list(np.shape(v)), "dtype": str(np.asarray(v).dtype), "sha256": H.digest(np.asarray(v))} for k, v in arrays.items()}


def main():
    assert H.have_reference(), "needs /root/reference"
    H.build_reference()
    ref = H.Unit("ref")
    manifest = {"generator": "tests/golden/generate.py", "source": "compiled reference (oracle/_ref), gcc -O2 -ffp-contract=off, glibc 2.35",
                "frame_seed": hex(scenes.FRAME_SEED), "files": {}}

    # A2: camera-jitter PCG32
    seeds = np.array([0, 1, 0x5EED0001, 0xFFFFFFFF], np.uint32)
    manifest["files"]["pcg"] = save("pcg", seeds=seeds, floats=ref.pcg(seeds, 64))

    # A11: camera
    r = H.rng(21)
    d = scenes.cornell_box(1920, 1080)
    cam = scenes.camera_of(d)
    xy = np.stack([r.randint(0, 1920, 256), r.randint(0, 1080, 256)], axis=1).astype(np.uint32)
    xy[:4] = [[0, 0], [1919, 0], [0, 1079], [1919, 1079]]
    rr = r.uniform(0, 1, size=(256, 2)).astype(np.float32)
    rr[:2] = [[0, 0], [1, 1]]
    manifest["files"]["camera"] = save("camera", xy=xy, r=rr, jitter=np.float32(0.5), dirs=ref.camera_dirs(cam, 1920, 1080, xy, 0.5, rr))
    cam2 = api.TerraCamera(); cam2.position = api.f3((0.3, 1.2, -2.0)); cam2.direction = api.f3((0.2, -0.1, 1.0)); cam2.up = api.f3((0.05, 1.0, 0.0)); cam2.fov = 60.0
    manifest["files"]["camera_tilted"] = save("camera_tilted", xy=xy, r=rr, jitter=np.float32(0.25), dirs=ref.camera_dirs(cam2, 1920, 1080, xy, 0.25, rr))

    # A6: slab test
    o, dd, boxes = H.aabb_cases()
    hit, tmin, tmax = ref.ray_aabb(o, dd, boxes)
    manifest["files"]["ray_aabb"] = save("ray_aabb", o=o, d=dd, boxes=boxes, hit=hit, tmin=tmin, tmax=tmax)

    # A7 / A7': triangle tests
    o, dd, tris = H.watertight_cases()
    hit, out = ref.watertight(o, dd, tris)
    manifest["files"]["watertight"] = save("watertight", o=o, d=dd, tris=tris, hit=hit, out=out)
    hit, out = ref.moller_trumbore(o, dd, tris)
    manifest["files"]["moller_trumbore"] = save("moller_trumbore", o=o, d=dd, tris=tris, hit=hit, out=out)

    # A16 + A5 + A4/A8 on Cornell-32
    L = ref.L
    dsc = scenes.cornell_box(256, 256, 4)
    scene = scenes.build_scene(L, dsc)
    manifest["files"]["bvh_cornell"] = save("bvh_cornell", nodes=ref.bvh_nodes(scene))
    o, dd = H.scene_rays(31, 4096)
    found, prim, point = ref.bvh_traverse(scene, o, dd)
    manifest["files"]["bvh_traverse"] = save("bvh_traverse", o=o, d=dd, found=found, prim=prim, point=point)
    o, dd = H.scene_rays(32, 2048)
    obj, tri, point, surf = ref.raycast(scene, o, dd)
    manifest["files"]["raycast"] = save("raycast", o=o, d=dd, obj=obj, tri=tri, point=point, surface=surf)
    L.scene_destroy(scene)

    # A12/A13: BSDF presets
    for kind_id, name in [(0, "diffuse"), (1, "phong")]:
        surf, e, wo = H.bsdf_cases(41 + kind_id, 2048, kind_id)
        wi, pdf, f, surf_after = ref.bsdf(kind_id, surf, e, wo)
        manifest["files"][f"bsdf_{name}"] = save(f"bsdf_{name}", surfaces=surf, e=e, wo=wo, wi=wi, pdf=pdf, f=f, pick=surf_after[:, 32])

    # A3/A9: terra_trace per primary ray, every integrator, diffuse and Phong scenes
    for sname, mk in [("cornell", scenes.cornell_box), ("phong", scenes.cornell_phong)]:
        for integ in range(7):
            dsc = mk(64, 64, 1, integrator=integ)
            scene = scenes.build_scene(L, dsc)
            o, dd = H.scene_rays(50 + integ, 1024)
            stateB, incB = H.stream_states(60 + integ, 1024)
            rad, calls = ref.trace(scene, o, dd, stateB, incB)
            manifest["files"][f"trace_{sname}_{integ}"] = save(f"trace_{sname}_{integ}", o=o, d=dd, stateB=stateB, incB=incB, radiance=rad, rand_calls=calls.astype(np.uint16))
            L.scene_destroy(scene)

    # A1: end-to-end. Config 1 (BASELINE.json configs[0]): Cornell, 256x256, 4 spp, Simple, fixed seed
    out = ref.render_pixels(scenes.cornell_box(256, 256, 4))
    manifest["files"]["render_config1"] = save("render_config1", acc=out["acc"], samples=np.int32(out["samples"][0, 0]), rand_calls=out["rand_calls"].astype(np.uint8))
    manifest["config1"] = {"pixels_sha256": H.digest(out["pixels"]), "pixels_fnv1a64": hex(H.fnv1a(out["pixels"])), "mean": float(out["pixels"].mean())}
    # small crops: other integrators / tonemaps / Phong / two accumulating passes
    small = {}
    for sname, mk in [("cornell", scenes.cornell_box), ("phong", scenes.cornell_phong)]:
        for integ in range(7):
            for tm in ([0, 1, 2, 3, 4] if (integ == 0 and sname == "cornell") else [0]):
                o2 = ref.render_pixels(mk(48, 32, 3, integrator=integ, tonemap=tm), passes=2)
                key = f"{sname}_i{integ}_t{tm}"
                small[key + "_pixels"] = o2["pixels"]; small[key + "_acc"] = o2["acc"]; small[key + "_calls"] = o2["rand_calls"].astype(np.uint16)
    # stratified sampling rounds spp up (reference src/Terra.c:519-527)
    o3 = ref.render_pixels(scenes.cornell_box(16, 16, 5, sampling=api.kTerraSamplingMethodStratified, strata=2))
    small["stratified_pixels"] = o3["pixels"]; small["stratified_samples"] = o3["samples"]
    # a tile of a larger frame (tile offsets, non-square aspect)
    o4 = ref.render_pixels(scenes.cornell_box(160, 90, 2), rect=(48, 16, 64, 32))
    small["tile_pixels"] = o4["pixels"]; small["tile_samples"] = o4["samples"]
    manifest["files"]["render_small"] = save("render_small", **small)

    # config 3 geometry: the ~100k-triangle hall (deep reference-tree traversal), small frames
    hall = {}
    for integ, (w, h, spp) in {0: (160, 90, 2), 1: (64, 36, 1)}.items():
        o5 = ref.render_pixels(scenes.sponza_hall(w, h, spp, integrator=integ))
        hall[f"i{integ}_pixels"] = o5["pixels"]; hall[f"i{integ}_calls"] = o5["rand_calls"].astype(np.uint16)
    dh = scenes.sponza_hall(64, 36, 1)
    sc = scenes.build_scene(L, dh)
    nodes = ref.bvh_nodes(sc)
    hall["bvh_sha256"] = np.frombuffer(bytes.fromhex(H.digest(nodes)), dtype=np.uint8)
    hall["bvh_nodes"] = np.int64(len(nodes))
    o, dd = H.scene_rays(71, 512, box=((-9.5, 0.3, -4.5), (9.5, 7.5, 4.5)))
    found, prim, point = ref.bvh_traverse(sc, o, dd)
    hall["trav_found"] = found; hall["trav_prim"] = prim; hall["trav_point"] = point
    L.scene_destroy(sc)
    manifest["files"]["render_hall"] = save("render_hall", **hall)

    # the same hall with every coordinate (scene and camera) x 100: outside the +-13-unit range in which the 1e-4 box margins provably exceed rounding error;
    # the product's automatic mode renders it with the fast tree and the reference's reachability replayed -- this is what it has to reproduce
    from tools.scaled_hall import scaled
    hall100 = {}
    for integ, (w, h, spp) in {0: (160, 90, 2), 1: (64, 36, 1)}.items():
        o5 = ref.render_pixels(scaled(scenes.sponza_hall(w, h, spp, integrator=integ), 100.0))
        hall100[f"i{integ}_pixels"] = o5["pixels"]; hall100[f"i{integ}_calls"] = o5["rand_calls"].astype(np.uint16)
    manifest["files"]["render_hall_x100"] = save("render_hall_x100", **hall100)

    # textured attributes (byte/float textures, point/bilinear, wrap/clamp, textured emissive)
    tex = {}
    for integ in (0, 1, 2):
        o6 = ref.render_pixels(scenes.cornell_textured(64, 48, 3, integrator=integ), passes=2)
        tex[f"i{integ}_pixels"] = o6["pixels"]; tex[f"i{integ}_calls"] = o6["rand_calls"].astype(np.uint16)
    manifest["files"]["render_textured"] = save("render_textured", **tex)

    # SURVEY 8f N4 (unit level): stratified / Halton samplers, 1D / 2D distributions (reference src/Terra.c:703-846)
    smp = {}
    seeds = np.array([0, 7, 0x5EED0001, 0xFFFFFFFF], np.uint32)
    for strata, samples in ((1, 1), (2, 3), (4, 16), (7, 2)):
        smp[f"strat_{strata}_{samples}"] = ref.stratified(seeds, strata, samples, strata * strata * samples)
    smp["seeds"] = seeds
    smp["halton_0"] = ref.halton(0, 4096)
    smp["halton_far"] = ref.halton(2 ** 30, 512)
    tables, e, t2, e12 = H.sampler_cases()
    smp["e"] = e; smp["e12"] = e12
    for name, f in tables.items():
        o7 = ref.distribution_1d(f, e)
        smp[f"d1_{name}_f"] = f
        for k, v in o7.items():
            smp[f"d1_{name}_{k}"] = v
    for name, f in t2.items():
        o8 = ref.distribution_2d(f, e12)
        smp[f"d2_{name}_f"] = f
        for k, v in o8.items():
            smp[f"d2_{name}_{k}"] = v
    manifest["files"]["samplers"] = save("samplers", **smp)

    (HERE / "manifest.json").write_text(json.dumps(manifest, indent=1))
    total = sum(p.stat().st_size for p in HERE.glob("*.npz"))
    print(f"wrote {len(manifest['files'])} fixtures, {total / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
