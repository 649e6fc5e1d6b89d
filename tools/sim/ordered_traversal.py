"""Statistics of traversal variants on the Cornell reference tree (dev tool): nodes popped / triangle tests per ray for
  A  the reference's traversal (everything tested),
  B  + leaf-box cull,
  C  ordered (near child first), subtrees and leaves culled against the closest hit so far.
float64 geometry: statistics only."""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import harness as H
from terra_amd import api, scenes

H.build_oracle(); L = H.lib("orc")
d = scenes.cornell_box(1920, 1080, 4); scene = scenes.build_scene(L, d)
nodes = H.Unit("orc").bvh_nodes(scene)          # (n,16) uint32: aabb0 min,max aabb1 min,max index[2] type[2]
nf = nodes.view(np.float32)
tris = []
for o in d.objects: tris.append(o.triangles)
obj_first = np.cumsum([0] + [len(o.triangles) for o in d.objects])
alltris = np.concatenate(tris).astype(np.float64)

def slab(o, inv, lo, hi):
    t1 = (lo - o) * inv; t2 = (hi - o) * inv
    tmin = np.max(np.minimum(t1, t2)); tmax = np.min(np.maximum(t1, t2))
    return tmax > max(tmin, 0.0), max(tmin, 0.0)

def tri_hit(o, dd, T):
    e1 = T[1] - T[0]; e2 = T[2] - T[0]; h = np.cross(dd, e2); a = e1 @ h
    if abs(a) < 1e-12: return None
    f = 1 / a; s = o - T[0]; u = f * (s @ h)
    if u < 0 or u > 1: return None
    q = np.cross(s, e1); v = f * (dd @ q)
    if v < 0 or u + v > 1: return None
    t = f * (e2 @ q)
    return t if t >= 0 else None

def child(n, c):
    lo = nf[n, 6 * c:6 * c + 3].astype(np.float64); hi = nf[n, 6 * c + 3:6 * c + 6].astype(np.float64)
    idx = int(nodes[n, 12 + c]); typ = int(nodes[n, 14 + c].view(np.int32)) if False else int(np.int32(nodes[n, 14 + c]))
    return lo, hi, idx, typ

def trav(o, dd, mode):
    inv = 1.0 / np.where(dd == 0, 1e-30, dd)
    best = np.inf; nn = 0; nt = 0
    if mode in "AB":
        st = [0]
        while st:
            n = st.pop(); nn += 1
            for c in (0, 1):
                lo, hi, idx, typ = child(n, c)
                h, te = slab(o, inv, lo, hi)
                if typ == -1:
                    if h: st.append(idx)
                elif typ == 1:
                    if mode == "A" or h:
                        nt += 1
                        t = tri_hit(o, dd, alltris[obj_first[idx & 0xff] + (idx >> 8)])
                        if t is not None and t < best: best = t
    else:
        st = [(0, -1)]         # (node or ~leaf, entry)
        while st:
            n, te = st.pop()
            if te > best: continue
            if n < 0:
                idx = ~n; nt += 1
                t = tri_hit(o, dd, alltris[obj_first[idx & 0xff] + (idx >> 8)])
                if t is not None and t < best: best = t
                continue
            nn += 1
            cand = []
            for c in (0, 1):
                lo, hi, idx, typ = child(n, c)
                h, t0 = slab(o, inv, lo, hi)
                if h and t0 <= best: cand.append((t0, idx if typ == -1 else ~idx))
            cand.sort(key=lambda x: -x[0])          # far first, near popped first
            for t0, x in cand: st.append((x, t0))
    return nn, nt, best

rng = np.random.RandomState(3)
camf = L.fn("orc_camera_sample", api.TerraFloat3, [C.POINTER(api.TerraCamera), C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_float, C.c_float, C.c_float])
cam = scenes.camera_of(d)
F3P = C.POINTER(api.TerraFloat3)
raycast = L.fn("orc_raycast", C.c_int, [C.c_void_p, F3P, F3P, C.POINTER(api.TerraShadingSurface), F3P, C.POINTER(C.c_int)])
stats = {m: [] for m in "ABC"}
for it in range(600):
    px, py = rng.randint(450, 1470), rng.randint(100, 980)
    dv = camf(C.byref(cam), 1920, 1080, px, py, 0.5, rng.rand(), rng.rand()); o = np.array([0, 1, -3.4]); dd = np.array(dv.tuple())
    for bounce in range(6):
        res = [trav(o + dd * 1e-3, dd, m) for m in "ABC"]
        assert abs(res[0][2] - res[1][2]) < 1e-9 or (np.isinf(res[0][2]) and np.isinf(res[1][2])), res
        assert abs(res[0][2] - res[2][2]) < 1e-9 or (np.isinf(res[0][2]) and np.isinf(res[2][2])), res
        for m, r in zip("ABC", res): stats[m].append((r[0], r[1], bounce))
        surf = api.TerraShadingSurface(); p = api.TerraFloat3(); t = C.c_int(0)
        obj = raycast(scene, C.byref(api.TerraFloat3(*map(float, o))), C.byref(api.TerraFloat3(*map(float, dd))), C.byref(surf), C.byref(p), C.byref(t))
        if obj < 0: break
        n = np.array([surf.normal.x, surf.normal.y, surf.normal.z]); P = np.array(p.tuple())
        e1, e2 = rng.rand(), rng.rand(); r = np.sqrt(e1); th = 2 * np.pi * e2
        a = np.array([1, 0, 0]) if abs(n[0]) < 0.9 else np.array([0, 1, 0]); tg = np.cross(n, a); tg /= np.linalg.norm(tg); bt = np.cross(n, tg)
        dd = r * np.cos(th) * tg + r * np.sin(th) * bt + np.sqrt(max(0, 1 - e1)) * n
        o = P + n * 1e-4
        if rng.rand() > 0.7: break
for m in "ABC":
    a = np.array(stats[m]); sec = a[a[:, 2] > 0]; pri = a[a[:, 2] == 0]
    def wm(x):      # mean of the max over random groups of 64
        x = x[rng.permutation(len(x))][: len(x) // 64 * 64].reshape(-1, 64); return x.max(1).mean()
    print(m, "all rays: nodes %.2f tris %.2f | primary nodes %.2f tris %.2f | secondary nodes %.2f (max64 %.1f) tris %.2f (max64 %.1f)" % (a[:, 0].mean(), a[:, 1].mean(), pri[:, 0].mean(), pri[:, 1].mean(), sec[:, 0].mean(), wm(sec[:, 0]), sec[:, 1].mean(), wm(sec[:, 1])))
