"""Throughput against scene size: the procedural hall at rising tessellation (97k .. 650k triangles), fast tree built on the
host (binned SAH) and on the device (LBVH), Simple integrator, 1920x1080 x 16 spp; also the commit time and its phases
(terra_amd_set_commit_timing(1) prints them on stderr). One GPU."""
import torch  # first
import ctypes as C, os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from terra_amd import api, runtime, scenes
lib = runtime.load()
lib.set_commit_timing(1)
for detail, builder in [(dt, b) for dt in (1.0, 2.0, 3.2) for b in (0, 1)]:
    d = scenes.sponza_hall(1920, 1080, 16, detail=detail)
    t = time.perf_counter(); s = scenes.build_scene(lib, d, tree_mode=2, tree_builder=builder); commit = time.perf_counter() - t
    assert runtime.last_error() == "", runtime.last_error()
    fb = runtime.DeviceFramebuffer(d.width, d.height); cam = scenes.camera_of(d)
    runtime.render_device(lib, cam, s, fb); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); runtime.render_device(lib, cam, s, fb); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1); st = runtime.Stats(); lib.get_stats(s, C.byref(st)); st = st.as_dict()
    print(json.dumps({"triangles": d.triangle_count, "builder": "device LBVH" if builder else "host SAH", "commit_ms": round(commit * 1e3), "render_ms": round(ms, 2), "Msamples/s": round(d.width * d.height * d.spp / ms / 1e3, 1),
                      "nodes_per_ray": round(st["nodes"] / st["rays"], 1), "tri_tests_per_ray": round(st["tri_tests"] / st["rays"], 1)}), flush=True)
    lib.scene_destroy(s)
