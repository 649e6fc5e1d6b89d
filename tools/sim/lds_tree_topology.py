"""Would a 3-axis SAH tree staged in LDS beat the reference tree on the Cornell box? (CPU model, numpy)
Both trees are traversed the way the LDS-resident kernel does it: no ordering, no culling against the closest hit, a child is entered iff the ray passes its box; a leaf child
costs a triangle test iff the ray passes its (+-1e-4) box (the leaf-box cull). Counts node steps (pops of an inner node) and triangle tests per ray for camera rays and for
bounce-like rays (origin on a random surface point, cosine-distributed direction)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from terra_amd import scenes

d = scenes.cornell_box(1920, 1080, 1)
tris = np.concatenate([np.asarray(o.triangles, np.float64) for o in d.objects])          # (n, 3, 3)
n = len(tris)
lo = tris.min(axis=1) - 1e-4; hi = tris.max(axis=1) + 1e-4

def area(a, b):
    e = np.maximum(b - a, 0); return e[0] * e[1] + e[1] * e[2] + e[2] * e[0]

def build_sah(idx):
    """returns nested tuple: ('leaf', i) or ('node', left, right, (lo, hi) of each child)"""
    if len(idx) == 1:
        return ('leaf', idx[0])
    best = None
    for ax in range(3):
        order = sorted(idx, key=lambda i: (lo[i, ax] + hi[i, ax]))
        for k in range(1, len(order)):
            L, R = order[:k], order[k:]
            c = area(lo[L].min(0), hi[L].max(0)) * len(L) + area(lo[R].min(0), hi[R].max(0)) * len(R)
            if best is None or c < best[0]:
                best = (c, L, R)
    _, L, R = best
    return ('node', build_sah(L), build_sah(R))

def build_ref(idx):
    """the reference's builder in outline (src/TerraBVH.c:128-244): sort by box centre x (descending, stable), sweep SAH along that order only"""
    if len(idx) == 1:
        return ('leaf', idx[0])
    order = sorted(idx, key=lambda i: -(lo[i, 0] + hi[i, 0]))
    best = None
    for k in range(1, len(order)):
        L, R = order[:k], order[k:]
        c = area(lo[L].min(0), hi[L].max(0)) * len(L) + area(lo[R].min(0), hi[R].max(0)) * len(R)
        if best is None or c < best[0]:
            best = (c, L, R)
    _, L, R = best
    return ('node', build_ref(L), build_ref(R))

def bounds(t):
    if t[0] == 'leaf':
        return lo[t[1]], hi[t[1]]
    a, b = bounds(t[1]), bounds(t[2])
    return np.minimum(a[0], b[0]), np.maximum(a[1], b[1])

def slab(o, inv, b):
    t1 = (b[0] - o) * inv; t2 = (b[1] - o) * inv
    tmin = np.minimum(t1, t2).max(); tmax = np.maximum(t1, t2).min()
    return tmax > max(tmin, 0.0)

def traverse(t, o, inv):
    nodes = tests = 0
    stack = [t]
    while stack:
        nd = stack.pop(); nodes += 1
        for ch in (nd[1], nd[2]):
            if slab(o, inv, bounds_cache[id(ch)]):
                if ch[0] == 'leaf': tests += 1
                else: stack.append(ch)
    return nodes, tests

def cache(t, c):
    c[id(t)] = bounds(t)
    if t[0] == 'node': cache(t[1], c); cache(t[2], c)

r = np.random.RandomState(3)
cam = np.array(d.camera_position, np.float64)
def camera_rays(m):
    out = []
    for _ in range(m):
        x, y = r.uniform(-1, 1), r.uniform(-1, 1)
        t = np.tan(np.radians(d.camera_fov) / 2)
        dr = np.array([x * t * 16 / 9, y * t, 1.0]); dr /= np.linalg.norm(dr)
        out.append((cam, dr))
    return out
def bounce_rays(m):
    out = []
    a = np.linalg.norm(np.cross(tris[:, 1] - tris[:, 0], tris[:, 2] - tris[:, 0]), axis=1)
    for _ in range(m):
        i = r.choice(n, p=a / a.sum()); u, v = r.uniform(), r.uniform()
        if u + v > 1: u, v = 1 - u, 1 - v
        p = tris[i, 0] + u * (tris[i, 1] - tris[i, 0]) + v * (tris[i, 2] - tris[i, 0])
        nrm = np.cross(tris[i, 1] - tris[i, 0], tris[i, 2] - tris[i, 0]); nrm /= np.linalg.norm(nrm)
        if r.uniform() < 0.5: nrm = -nrm
        w = r.normal(size=3); w /= np.linalg.norm(w)
        if w @ nrm < 0: w = -w
        out.append((p + nrm * 1e-3, w))
    return out

for name, builder in (("reference-like (x only)", build_ref), ("3-axis SAH", build_sah)):
    tree = builder(list(range(n)))
    bounds_cache = {}; cache(tree, bounds_cache)
    for kind, rays in (("camera", camera_rays(3000)), ("bounce", bounce_rays(3000))):
        res = np.array([traverse(tree, o, 1.0 / np.where(dd == 0, 1e-30, dd)) for o, dd in rays])
        print(f"{name:26s} {kind:7s} rays: node steps {res[:, 0].mean():6.2f} (p95 {np.percentile(res[:, 0], 95):4.0f}, max {res[:, 0].max():3d})   triangle tests {res[:, 1].mean():5.2f} (max {res[:, 1].max()})")
