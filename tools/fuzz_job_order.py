"""Randomised run for the job order (GPU box; not a unit test): small random triangle soups (LDS-resident scenes) seen through random cameras on frames large enough for the
default job order (>= 256 pixel blocks), random integrators, sample splits (incl. the automatic one), rectangles and shards. The device frame with the job order (default) must
equal the oracle's (device-twin math) bit for bit, and the frame without it.
    python tools/fuzz_job_order.py [iterations] [seed]"""
import torch  # first
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from terra_amd import api, runtime, scenes

n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
lib = runtime.load()
orc = api.TerraLib(os.path.join(ROOT, "oracle", "liboracle.so"), "orc_")
orc.fn("orc_set_math_mode", None, [C.c_int])(1)
orp = orc.fn("orc_render_pixels", None, [C.POINTER(api.TerraCamera), C.c_void_p, C.POINTER(api.TerraFramebuffer)] + [C.c_size_t] * 4 + [C.c_uint64, C.c_void_p])


def soup(n_tris, n_objects):
    objs = []
    per = max(1, n_tris // n_objects); left = n_tris
    for k in range(n_objects):
        n = per if k < n_objects - 1 else left
        if n <= 0: break
        left -= n
        c = rs.uniform(-1.5, 1.5, size=(n, 1, 3)); tris = (c + rs.uniform(-0.7, 0.7, size=(n, 3, 3))).astype(np.float32)
        nrm = np.cross(tris[:, 1] - tris[:, 0], tris[:, 2] - tris[:, 0]); nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-20)
        nrm = np.repeat(nrm[:, None, :], 3, axis=1).astype(np.float32)
        m = scenes.Material(kind=str(rs.choice(["diffuse", "diffuse", "phong"])), albedo=tuple(rs.uniform(0.2, 0.9, 3)), emissive=(4.0, 3.0, 2.0) if k == 0 else (0.0, 0.0, 0.0),
                            specular_color=tuple(rs.uniform(0.1, 0.9, 3)), specular_intensity=float(rs.choice([1.0, 8.0, 30.5])))
        objs.append(scenes.ObjectDesc(tris, nrm, rs.uniform(0, 1, size=(n, 3, 2)).astype(np.float32), m))
    return objs


def bits_equal(a, b):
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    na, nb = np.isnan(a), np.isnan(b)
    return np.array_equal(na, nb) and np.array_equal(a.view(np.uint32)[~na], b.view(np.uint32)[~nb])


bad = 0; ordered = 0
for it in range(n_iter):
    W, H = int(rs.randint(260, 520)), int(rs.randint(250, 400))
    integ = int(rs.choice([0, 0, 1, 2, 5])); split = int(rs.choice([0, 1, 2, 4])); spp = (split or 1) * int(rs.randint(1, 3))
    dist = float(rs.uniform(3.0, 12.0)); off = rs.uniform(-1.5, 1.5, size=2)
    d = scenes.SceneDesc(objects=soup(int(rs.choice([3, 12, 30, 60])), int(rs.randint(1, 4))), width=W, height=H, spp=spp, bounces=int(rs.randint(0, 5)), integrator=integ,
                         camera_position=(float(off[0]), float(off[1]), -dist), camera_fov=float(rs.uniform(25, 80)), tonemap=int(rs.randint(0, 5)), environment=(0.2, 0.3, 0.4),
                         environment_lighting=bool(rs.randint(2)))
    cam = scenes.camera_of(d)
    mode = int(rs.randint(3))          # 0 whole frame, 1 rectangle, 2 shards rendered one after the other
    rect = (0, 0, W, H) if mode != 1 else (int(rs.randint(0, 40)), int(rs.randint(0, 40)), W - int(rs.randint(40, 80)), H - int(rs.randint(40, 80)))
    world = int(rs.choice([2, 3])) if mode == 2 else 1
    outs = []
    used = None
    for order in (1, 0):
        lib.clear_error()
        s = scenes.build_scene(lib, d, counters=bool(rs.randint(2)))
        runtime.check(lib.set_job_order(s, order)); runtime.check(lib.set_sample_split(s, split))
        fb = runtime.DeviceFramebuffer(W, H)
        if mode == 2:
            for r in range(world):
                runtime.render_device_sharded(lib, cam, s, fb, 64, r, world)
        else:
            runtime.render_device(lib, cam, s, fb, rect)
        torch.cuda.synchronize()
        outs.append((fb.results_host()["acc"].copy(), fb.pixels_host().copy()))
        lib.scene_destroy(s)
    # the oracle: the split the device used is its own choice when split == 0 -- ask the library (these scenes are LDS-resident: the job-ordered rule)
    blocks_ok = True
    eff = split or lib.auto_sample_split(rect[2], rect[3], 64, world, spp, 1)
    while eff > 1 and spp % eff: eff >>= 1
    so = scenes.build_scene(orc, scenes.SceneDesc(**{**d.__dict__, "spp": spp // eff})); fo = api.Framebuffer(orc, W, H)
    for _ in range(eff):
        orp(C.byref(cam), so, C.byref(fo.fb), rect[0], rect[1], rect[2], rect[3], scenes.FRAME_SEED, None)
    x, y, w, h = rect
    crop = lambda a: a[y:y + h, x:x + w]
    ok_orc = bits_equal(crop(outs[0][0]), crop(fo.results["acc"])) and bits_equal(crop(outs[0][1]), crop(fo.pixels))
    ok_off = bits_equal(outs[0][0], outs[1][0]) and bits_equal(outs[0][1], outs[1][1])
    empty = float((np.abs(crop(outs[0][0])).sum(axis=-1) == 0).mean())
    ordered += int(0.02 < empty < 0.98)
    if not (ok_orc and ok_off):
        bad += 1; print("MISMATCH", dict(it=it, W=W, H=H, integ=integ, split=split, eff=eff, spp=spp, bounces=d.bounces, mode=mode, rect=rect, world=world, vs_oracle=ok_orc, vs_unordered=ok_off))
    fo.destroy(); orc.scene_destroy(so)
    if (it + 1) % 50 == 0: print(f"  {it + 1} cases, {bad} mismatches so far", file=sys.stderr, flush=True)
print(f"{n_iter} cases, {bad} mismatches, {ordered} frames with both hit and empty regions, last error: '{runtime.last_error()}'")
sys.exit(1 if bad else 0)
