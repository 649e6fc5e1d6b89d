"""The in-process multi-GPU path of the C-ABI (terra_amd_set_devices + terra_amd_render_multi: one process, tiles dealt to the devices, ONE RCCL gather issued
by the library, one copy to the host) timed on every visible device count 1, 2, 4, ... -- a child process of bench.py (so that nothing it does can take the bench
line down with it), or stand-alone:

    python tools/multi_device_bench.py [--workload cornell_1080p_512spp] [--steps 3] [--max-devices 8]

Prints one JSON line: per device count the wall time of terra_amd_render_multi() on a pinned host framebuffer (PCIe included: 16 B per pixel up to every device,
28 B per pixel down from the primary), the rate, whether the frame equals the one-device frame bit for bit, and what terra_amd_multi_info() reports.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cornell_1080p_512spp")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--max-devices", type=int, default=8)
    ap.add_argument("--sample-split", type=int, default=0, help="terra_amd_set_sample_split (0: the library's automatic choice per launch, as bench.py's default)")
    ap.add_argument("--job-order", type=int, default=-1, help="terra_amd_set_job_order (A/B; -1: the library's default)")
    a = ap.parse_args()
    import numpy as np
    import torch  # noqa: F401  first: the library binds to the HIP runtime torch loads
    import bench
    from terra_amd import api, runtime, scenes
    lib = runtime.load()
    ndev = min(lib.device_count(), a.max_devices)
    d = bench.workload(a.workload)
    out = {"workload": a.workload, "visible_devices": lib.device_count(), "steps": a.steps, "runs": []}
    first = None
    n = 1
    while n <= ndev:
        devs = (C.c_int * n)(*range(n))
        runtime.check(lib.set_devices(devs, n), "terra_amd_set_devices")
        lib.clear_error()
        scene = scenes.build_scene(lib, d, counters=False)
        if runtime.last_error():
            out["runs"].append({"devices": n, "error": runtime.last_error()}); break
        runtime.check(lib.set_sample_split(scene, a.sample_split))
        if a.job_order >= 0:
            runtime.check(lib.set_job_order(scene, a.job_order))
        fb = api.Framebuffer(lib, d.width, d.height); cam = scenes.camera_of(d)
        rc = lib.render_multi(C.byref(cam), scene, C.byref(fb.fb), 0, 0, d.width, d.height, 64)          # warm-up: replicas' first launch, communicator, staging buffers
        if rc != 0:
            out["runs"].append({"devices": n, "error": runtime.last_error()}); break
        frame1 = fb.results["acc"].copy()
        t = time.perf_counter()
        for _ in range(a.steps):
            lib.render_multi(C.byref(cam), scene, C.byref(fb.fb), 0, 0, d.width, d.height, 64)
        dt = (time.perf_counter() - t) / a.steps
        info = runtime.MultiInfo(); runtime.check(lib.multi_info(scene, C.byref(info)))
        if first is None:
            first = frame1
        out["runs"].append({"devices": n, "ms_per_step": round(dt * 1e3, 3), "value": round(d.width * d.height * d.spp / dt / 1e6, 2), "unit": "Msamples/s",
                            "equals_one_device_frame": bool(np.array_equal(frame1.view(np.uint32), first.view(np.uint32))), "gathers": int(info.gathers),
                            "gather_bytes": int(info.last_gather_bytes), "communicator_ranks": int(info.communicator_ranks), "rccl_version": int(info.rccl_version),
                            "commit_ms": round(scenes.LAST_COMMIT_MS, 2)})
        fb.destroy(); lib.scene_destroy(scene)
        n *= 2
    lib.set_devices(None, 0)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
