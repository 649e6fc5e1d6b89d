"""What one terra_render() call on a tile costs from a single thread (dev tool, GPU box): at 1 spp the kernel is negligible, so the time is the
call's fixed cost (upload of the running sums, launch, resolve, two downloads, the wait); at 512 spp the tile's own render is added.
    python3 tools/tile_call_overhead.py"""
import torch, ctypes as C, sys, time, os
sys.path.insert(0, os.getcwd())
from terra_amd import api, runtime, scenes
lib = runtime.load()
for spp in (1, 512):
    d = scenes.cornell_box(1920, 1080, spp)
    scene = scenes.build_scene(lib, d, counters=False); cam = scenes.camera_of(d)
    fb = api.Framebuffer(lib, d.width, d.height)
    runtime.check(lib.set_sample_split(scene, 0))
    for size in (128, 256):
        lib.render(C.byref(cam), scene, C.byref(fb.fb), 0, 0, size, size)
        n = 50; t = time.perf_counter()
        for i in range(n):
            lib.render(C.byref(cam), scene, C.byref(fb.fb), (i % 7) * size, 0, size, size)
        dt = (time.perf_counter() - t) / n
        print(f"spp {spp} tile {size}: {dt*1e6:.0f} us per terra_render() call from one thread", runtime.last_error())
