"""Per-ray work traces of the Cornell box for the lane-scheduling simulator (dev tool).
For 8x8 pixel packets at several frame positions: every lane's sequence of rays over `spp` samples, each ray =
(nodes popped, triangle tests, hit, is_camera). Paths are diffuse bounces with Russian roulette like terra_trace
(not bit-exact -- statistics only); traversal work comes from the oracle's counters around orc_raycast."""
import ctypes as C, sys, os, pickle
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import harness as H
from terra_amd import api, scenes

class Ctr(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("rays", "nodes", "box_tests", "tri_tests", "hits", "samples", "rand_calls", "attr_fetches")]

def main(spp=64, out="/tmp/cornell_traces.pkl"):
    H.build_oracle(); L = H.lib("orc")
    d = scenes.cornell_box(1920, 1080, spp); scene = scenes.build_scene(L, d); cam = scenes.camera_of(d)
    F3P = C.POINTER(api.TerraFloat3)
    raycast = L.fn("orc_raycast", C.c_int, [C.c_void_p, F3P, F3P, C.POINTER(api.TerraShadingSurface), F3P, C.POINTER(C.c_int)])
    reset = L.fn("orc_counters_reset", None, []); get = L.fn("orc_counters_get", None, [C.POINTER(Ctr)])
    camf = L.fn("orc_camera_sample", api.TerraFloat3, [C.POINTER(api.TerraCamera), C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_float, C.c_float, C.c_float])
    rng = np.random.RandomState(5)
    packets = [(x, y) for y in (200, 420, 540, 700, 900) for x in (200, 600, 960, 1300, 1700)]
    albedo = {0: (.73, .73, .73), 1: (.65, .05, .05), 2: (.12, .45, .15), 3: (.73,) * 3, 4: (.73,) * 3, 5: (.73,) * 3}
    traces = {}
    for (px0, py0) in packets:
        lanes = []
        for lane in range(64):
            px, py = px0 + (lane & 7), py0 + (lane >> 3)
            seq = []
            for s in range(spp):
                r1, r2 = rng.rand(), rng.rand()
                dv = camf(C.byref(cam), 1920, 1080, px, py, 0.5, r1, r2)
                o = np.array([0, 1, -3.4]); dd = np.array(dv.tuple())   # camera frame is identity for this scene
                thr = np.ones(3); is_cam = True
                for bounce in range(9):
                    surf = api.TerraShadingSurface(); p = api.TerraFloat3(); t = C.c_int(0)
                    reset()
                    obj = raycast(scene, C.byref(api.TerraFloat3(*map(float, o))), C.byref(api.TerraFloat3(*map(float, dd))), C.byref(surf), C.byref(p), C.byref(t))
                    c = Ctr(); get(C.byref(c))
                    hit = obj >= 0
                    seq.append((int(c.nodes), int(c.tri_tests), bool(hit), is_cam)); is_cam = False
                    if not hit: break
                    n = np.array([surf.normal.x, surf.normal.y, surf.normal.z]); P = np.array(p.tuple())
                    # cosine sample around n
                    e1, e2 = rng.rand(), rng.rand(); r = np.sqrt(e1); th = 2 * np.pi * e2
                    a = np.array([1, 0, 0]) if abs(n[0]) < 0.9 else np.array([0, 1, 0]); tg = np.cross(n, a); tg /= np.linalg.norm(tg); bt = np.cross(n, tg)
                    wi = r * np.cos(th) * tg + r * np.sin(th) * bt + np.sqrt(max(0, 1 - e1)) * n
                    thr = thr * np.array(albedo.get(obj, (.73,) * 3))
                    pr = thr.max()
                    if rng.rand() > pr: break
                    thr = thr / (pr + 1e-4)
                    o = P + n * 1e-4; dd = wi
            lanes.append(seq)
        traces[(px0, py0)] = lanes
        nr = sum(len(s) for s in lanes)
        print((px0, py0), "rays/sample %.2f nodes/ray %.2f tris/ray %.2f hit %.2f" % (nr / (64 * spp), sum(r[0] for s in lanes for r in s) / nr, sum(r[1] for s in lanes for r in s) / nr, sum(r[2] for s in lanes for r in s) / nr), flush=True)
    pickle.dump(traces, open(out, "wb"))

if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 64)
