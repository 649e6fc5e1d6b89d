"""Whole-frame comparison of two traversal modes of the product on one of bench.py's workloads (dev tool, GPU):
renders the frame in mode A and mode B (terra_amd_set_tree_mode: 0 replica, 1 fast tree, 2 automatic), reports the
pixels whose sums differ in any bit, the work counters of both launches and the kernel time.

    python tools/compare_modes.py --workload cornell_1080p_512spp --a 0 --b 1 [--split 8] [--spp N] [--integrator direct]
"""
import argparse, ctypes as C, os, sys, time
import torch  # noqa: F401  first: the library binds to the HIP runtime torch loads
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from terra_amd import runtime, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="cornell_1080p_512spp")
ap.add_argument("--a", type=int, default=0); ap.add_argument("--b", type=int, default=1)
ap.add_argument("--split", type=int, default=8); ap.add_argument("--spp", type=int, default=0)
ap.add_argument("--integrator", default="")
args = ap.parse_args()
L = runtime.load()
d = bench.workload(args.workload, args.spp)
if args.integrator:
    d.integrator = bench.INTEGRATORS[args.integrator]
out = {}
for mode in (args.a, args.b):
    scene = scenes.build_scene(L, d, tree_mode=mode)
    runtime.check(L.set_sample_split(scene, args.split))
    fb = runtime.DeviceFramebuffer(d.width, d.height); cam = scenes.camera_of(d)
    rc = torch.zeros(d.width * d.height, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize(); t = time.perf_counter()
    runtime.render_device(L, cam, scene, fb, None, rc)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    st = runtime.Stats(); runtime.check(L.get_stats(scene, C.byref(st)))
    ti = runtime.TraversalInfo(); runtime.check(L.traversal_info(scene, C.byref(ti)))
    res = fb.results_host()
    out[mode] = dict(acc=res["acc"].copy(), calls=rc.cpu().numpy().reshape(d.height, d.width), stats=st.as_dict(), note=ti.note.decode(), s=dt)
    print(f"mode {mode}: {dt * 1e3:.1f} ms (counting launch)  {ti.note.decode()}\n   {st.as_dict()}", flush=True)
    L.scene_destroy(scene)
a, b = out[args.a], out[args.b]
diff = (np.ascontiguousarray(a["acc"]).view(np.uint32) != np.ascontiguousarray(b["acc"]).view(np.uint32)).any(axis=-1)
cd = a["calls"] != b["calls"]
print(f"pixels whose sums differ: {int(diff.sum())} of {diff.size}; pixels whose draw counts differ: {int(cd.sum())}")
ys, xs = np.nonzero(diff | cd)
for y, x in list(zip(ys, xs))[:12]:
    print(f"   ({x},{y}) acc A {a['acc'][y, x]} B {b['acc'][y, x]} calls {a['calls'][y, x]} / {b['calls'][y, x]}")
sys.exit(1 if diff.any() or cd.any() else 0)
