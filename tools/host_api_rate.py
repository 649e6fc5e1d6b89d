"""PCIe-inclusive rate of the drop-in entry point terra_render() (host framebuffer) next to the
device-resident entry, same workload as bench.py. For DESIGN.md section 9; never bench.py's `value`."""
import torch  # noqa: F401
import ctypes as C, sys, time, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from terra_amd import api, runtime, scenes

lib = runtime.load()
out = {}
for spp in (512, 16):
    d = scenes.cornell_box(1920, 1080, spp)
    scene = scenes.build_scene(lib, d); cam = scenes.camera_of(d)
    fb = api.Framebuffer(lib, d.width, d.height)
    lib.render(C.byref(cam), scene, C.byref(fb.fb), 0, 0, d.width, d.height)          # warm-up (staging alloc)
    t = time.perf_counter(); n = 3
    for _ in range(n):
        lib.render(C.byref(cam), scene, C.byref(fb.fb), 0, 0, d.width, d.height)
    host = (time.perf_counter() - t) / n
    assert runtime.last_error() == ""
    dfb = runtime.DeviceFramebuffer(d.width, d.height)
    runtime.render_device(lib, cam, scene, dfb); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        runtime.render_device(lib, cam, scene, dfb)
    torch.cuda.synchronize(); dev = (time.perf_counter() - t) / n
    # tile-sized calls, as Satellite issues them (128-pixel tiles from one thread)
    t = time.perf_counter()
    for y in range(0, d.height, 128):
        for x in range(0, d.width, 128):
            lib.render(C.byref(cam), scene, C.byref(fb.fb), x, y, min(128, d.width - x), min(128, d.height - y))
    tiles = time.perf_counter() - t
    # the same tile calls with the automatic sample split, from one thread and from 8 worker threads (Satellite's pattern)
    import threading
    lib.set_sample_split(scene, 0)
    jobs = [(x, y, min(128, d.width - x), min(128, d.height - y)) for y in range(0, d.height, 128) for x in range(0, d.width, 128)]
    t = time.perf_counter()
    for j in jobs:
        lib.render(C.byref(cam), scene, C.byref(fb.fb), *j)
    tiles_auto = time.perf_counter() - t
    def worker(k):
        for j in jobs[k::8]:
            lib.render(C.byref(cam), scene, C.byref(fb.fb), *j)
    for split_mode, key in ((1, "tiles_8_threads_ms"), (0, "tiles_8_threads_auto_split_ms")):
        lib.set_sample_split(scene, split_mode)
        ths = [threading.Thread(target=worker, args=(k,)) for k in range(8)]
        t = time.perf_counter(); [th.start() for th in ths]; [th.join() for th in ths]
        out.setdefault(f"{spp}spp_threads", {})[key] = round((time.perf_counter() - t) * 1e3, 2)
    assert runtime.last_error() == ""
    s = d.width * d.height * spp / 1e6
    out[f"{spp}spp"] = {"terra_render_full_frame_ms": round(host * 1e3, 2), "Msamples/s": round(s / host, 1),
                        "device_resident_ms": round(dev * 1e3, 2), "device_Msamples/s": round(s / dev, 1),
                        "terra_render_135_tiles_ms": round(tiles * 1e3, 2), "tiles_Msamples/s": round(s / tiles, 1), "terra_render_135_tiles_auto_split_ms": round(tiles_auto * 1e3, 2),
                        "pcie_bytes_per_call": d.width * d.height * (16 + 16 + 12)}
    fb.destroy(); lib.scene_destroy(scene)
print(json.dumps(out, indent=1))
