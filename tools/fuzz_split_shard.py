"""Randomised self-consistency run on the GPU (not a unit test): random frame sizes, rectangles, tile sizes, shard
counts, sample splits and integrators; a split + sharded render must equal plain successive unsharded calls bit for bit.
    python tools/fuzz_split_shard.py [iterations] [seed]"""
import torch  # first
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from terra_amd import api, runtime, scenes

n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
lib = runtime.load()
makers = [scenes.cornell_box, scenes.cornell_phong, scenes.cornell_textured, lambda w, h, s, **k: scenes.cornell_spheres(w, h, s, **k)]
bad = 0
for it in range(n_iter):
    W, H = int(rs.randint(17, 200)), int(rs.randint(17, 150))
    split = int(rs.choice([2, 4, 8, 16])); chunk = int(rs.randint(1, 6)); spp = split * chunk
    integ = int(rs.choice([0, 1, 2, 4, 5])); mk = makers[rs.randint(len(makers))]
    tile = int(rs.choice([16, 32, 64, 128])); world = int(rs.randint(1, 6))
    x = int(rs.randint(0, W // 2)); y = int(rs.randint(0, H // 2)); w = int(rs.randint(1, W - x + 1)); h = int(rs.randint(1, H - y + 1))
    d = mk(W, H, spp, integrator=integ); cam = scenes.camera_of(d)
    a = runtime.DeviceFramebuffer(W, H); b = runtime.DeviceFramebuffer(W, H)
    sa = scenes.build_scene(lib, d); runtime.check(lib.set_sample_split(sa, split))
    for rank in range(world):
        runtime.check(lib.render_device_sharded(C.byref(cam), sa, a.pixels.data_ptr(), a.results.data_ptr(), W, H, x, y, w, h, tile, rank, world, None, None))
    d2 = mk(W, H, chunk, integrator=integ); sb = scenes.build_scene(lib, d2)
    for _ in range(split):
        runtime.check(lib.render_device(C.byref(cam), sb, b.pixels.data_ptr(), b.results.data_ptr(), W, H, x, y, w, h, None, None))
    torch.cuda.synchronize()
    ok = torch.equal(a.results, b.results) and torch.equal(a.pixels.view(torch.int32), b.pixels.view(torch.int32))
    if not ok:
        bad += 1; print("MISMATCH", dict(W=W, H=H, spp=spp, split=split, integ=integ, tile=tile, world=world, rect=(x, y, w, h), scene=d.name))
    lib.scene_destroy(sa); lib.scene_destroy(sb)
print(f"{n_iter} cases, {bad} mismatches, last error: '{runtime.last_error()}'")
sys.exit(1 if bad else 0)
