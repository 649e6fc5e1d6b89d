"""The hall with every coordinate (scene and camera) multiplied by a factor: outside the +-13-unit range of the containment proof the automatic mode
runs the fast tree with the reference's reachability replayed (DESIGN.md "Reachability"). Compares it with the replica traversal and times both.
    python tools/scaled_hall.py [--scale 100] [--spp 8] [--integrator 0]"""
import argparse, ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from terra_amd import scenes


def scaled(d, k):
    for o in d.objects:
        o.triangles = (np.asarray(o.triangles, np.float32) * np.float32(k)).astype(np.float32)
    d.camera_position = tuple(float(np.float32(c) * np.float32(k)) for c in d.camera_position)
    return d


if __name__ == "__main__":
    import torch  # before the library: libterra_amd.so must bind to the HIP runtime torch loaded
    ap = argparse.ArgumentParser(); ap.add_argument("--scale", type=float, default=100.0); ap.add_argument("--spp", type=int, default=8); ap.add_argument("--integrator", type=int, default=0)
    ap.add_argument("--width", type=int, default=1920); ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--shrink", type=float, default=0.0, help="terra_amd_debug_shrink_reference_boxes: the device's copy of the reference boxes shrunk by this much (gives the reachability replay work)")
    ap.add_argument("--builder", type=int, default=None, help="terra_amd_set_tree_builder: 0 host binned SAH, 1 device LBVH")
    a = ap.parse_args()
    from terra_amd import runtime
    L = runtime.load()
    outs = {}
    for mode in (2, 0):
        d = scaled(scenes.sponza_hall(a.width, a.height, a.spp, bounces=8, integrator=a.integrator), a.scale)
        L.clear_error(); s = scenes.build_scene(L, d, tree_mode=mode, debug_shrink=a.shrink or None, tree_builder=a.builder)
        ti = runtime.TraversalInfo(); runtime.check(L.traversal_info(s, C.byref(ti)))
        fb = runtime.DeviceFramebuffer(d.width, d.height); cam = scenes.camera_of(d)
        runtime.render_device(L, cam, s, fb); torch.cuda.synchronize(); fb.clear()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); runtime.render_device(L, cam, s, fb); e1.record(); torch.cuda.synchronize()
        st = runtime.Stats(); runtime.check(L.get_stats(s, C.byref(st))); st = st.as_dict()
        faults = L.fn("terra_amd_debug_faults", C.c_longlong, [C.c_void_p])(s)          # (self-checking builds: levels the replay mask had cleared that failed)
        ms = e0.elapsed_time(e1)
        print(f"scale {a.scale:g} tree mode {mode}: {ms:9.2f} ms  {d.width * d.height * a.spp / ms / 1e3:8.1f} Msamples/s  nodes/ray {st['nodes'] / max(1, st['rays']) / 2:.1f}  fast {ti.fast_tree}  check-build faults {faults}  note: {ti.note.decode()}", flush=True)
        outs[mode] = (fb.pixels_host().copy(), fb.results_host()["acc"].copy())
        L.scene_destroy(s)
    same = np.array_equal(outs[2][0].view(np.uint32), outs[0][0].view(np.uint32)) and np.array_equal(outs[2][1].view(np.uint32), outs[0][1].view(np.uint32))
    print("automatic == replica, bit for bit:", same)
    sys.exit(0 if same else 1)
