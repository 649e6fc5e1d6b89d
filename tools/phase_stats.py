"""Lane-occupancy study of the LDS-resident render kernel (dev tool, not a test).

Needs the instrumented build:  python -m terra_amd.build --variant ps -DTERRA_PHASE_STATS=1
Run on the GPU box:             TERRA_AMD_LIB=terra_amd/libterra_amd_ps.so python tools/phase_stats.py [--spp 512] [--split 8]

For each phase of the per-ray loop it prints the number of wave-level executions (x 64 = lane slots issued) and the
lanes that were active in them: slot utilisation = lanes / (64 x wave executions).
"""
import argparse
import ctypes as C
import os
import sys

import torch  # noqa: F401  first: the library binds to the HIP runtime torch loads

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from terra_amd import api, runtime, scenes  # noqa: E402

NAMES = ["ray_iter", "node_iter", "leaf_iter", "shade_iter", "cam_iter", "cam_lanes", "ray_lanes", "shade_lanes", "node_lanes", "leaf_lanes", "drain_iter", "top64", "top256", "top1024", "top4096"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spp", type=int, default=512)
    ap.add_argument("--split", type=int, default=8)
    ap.add_argument("--integrator", type=int, default=0)
    ap.add_argument("--scene", default="cornell")
    ap.add_argument("--tree", type=int, default=None, help="terra_amd_set_tree_mode")
    a = ap.parse_args()
    L = runtime.load()
    mk = {"cornell": scenes.cornell_box, "phong": scenes.cornell_phong, "hall": scenes.sponza_hall}[a.scene]
    d = mk(1920, 1080, a.spp, bounces=8, integrator=a.integrator)
    scene = scenes.build_scene(L, d, tree_mode=a.tree)
    runtime.check(L.set_sample_split(scene, a.split))
    fb = runtime.DeviceFramebuffer(d.width, d.height); cam = scenes.camera_of(d)
    runtime.render_device(L, cam, scene, fb)
    torch.cuda.synchronize()
    dbg = (C.c_ulonglong * 16)()
    f = L.fn("terra_amd_debug_counters", C.c_int, [C.c_void_p, C.c_void_p])
    runtime.check(f(scene, dbg))
    v = dict(zip(NAMES, list(dbg)))
    st = runtime.Stats(); runtime.check(L.get_stats(scene, C.byref(st))); st = st.as_dict()
    print(v); print(st)
    for ph, it, ln in (("cam", "cam_iter", "cam_lanes"), ("ray", "ray_iter", "ray_lanes"), ("node", "node_iter", "node_lanes"), ("leaf", "leaf_iter", "leaf_lanes"), ("shade", "shade_iter", "shade_lanes")):
        if v[it]:
            print(f"{ph:6s} wave execs {v[it]:14d}  lanes {v[ln]:16d}  utilisation {v[ln] / (64.0 * v[it]):.3f}   per ray-iter {v[it] / max(1, v['ray_iter']):.2f}")
    print(f"drains per ray-iter {v['drain_iter'] / max(1, v['ray_iter']):.3f}")
    if v["top4096"] and v["cam_lanes"] and not v["ray_iter"]:      # fast-tree launches (decoupled loop): the slots of the coupled loop hold the stack-depth histogram of the pushes
        n = v["cam_lanes"]
        print(f"fast tree stack: {n} pushes ({n / max(1, st['rays']):.2f} per ray); entries on the lane's stack after the push >= 4: {100.0 * v['shade_iter'] / n:.2f} %, >= 6: {100.0 * v['ray_lanes'] / n:.2f} %, "
              f">= 8: {100.0 * v['cam_iter'] / n:.3f} %, >= 10: {100.0 * v['drain_iter'] / n:.4f} %, >= 12: {100.0 * v['shade_lanes'] / n:.5f} %")
    if v["top4096"]:
        print("node visits served by a breadth-first prefix of the node array: " + ", ".join(f"first {k}: {100.0 * v['top' + str(k)] / st['nodes']:.1f} %" for k in (64, 256, 1024, 4096)) + f"  ({st['nodes'] / st['rays']:.1f} nodes per ray)")


if __name__ == "__main__":
    main()
