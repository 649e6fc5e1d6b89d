"""One frame of the reference client's call pattern -- terra_render() on 128-pixel tiles from 8 threads -- for a timeline trace:
    rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/tiles_trace -- python3 tools/host_tiles_trace.py
(then tools/host_tiles_trace.py --analyze gpurun_out/tiles_trace prints how the device's time was spent)."""
import sys, os, glob, csv
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def analyze(d):
    ks = []
    for f in glob.glob(f"{d}/**/*_kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40], r.get("Queue_Id", "")))
    cs = []
    for f in glob.glob(f"{d}/**/*_memory_copy_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            cs.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", "")))
    ks.sort()
    rk = [k for k in ks if "terra_render_kernel" in k[2]]
    # the last 135 render kernels = the timed frame
    rk = rk[-135:]
    t0, t1 = rk[0][0], max(k[1] for k in rk)
    print(f"frame: {len(rk)} render kernels over {(t1 - t0) / 1e6:.2f} ms; sum of kernel durations {sum(k[1] - k[0] for k in rk) / 1e6:.2f} ms; queues {sorted(set(k[3] for k in rk))}")
    ev = sorted([(k[0], 1) for k in rk] + [(k[1], -1) for k in rk])
    busy = {}; depth = 0; last = t0
    for t, s in ev:
        busy[depth] = busy.get(depth, 0) + (t - last); last = t; depth += s
    print("time with n render kernels in flight (ms):", {n: round(v / 1e6, 2) for n, v in sorted(busy.items())})
    durs = sorted((k[1] - k[0]) / 1e6 for k in rk)
    print(f"kernel duration min / median / max: {durs[0]:.3f} / {durs[len(durs) // 2]:.3f} / {durs[-1]:.3f} ms")
    inwin = [c for c in cs if c[0] >= t0 and c[1] <= t1 + 5e6]
    if inwin:
        print(f"copies in the window: {len(inwin)}, total {sum(c[1] - c[0] for c in inwin) / 1e6:.2f} ms, median {sorted(c[1] - c[0] for c in inwin)[len(inwin) // 2] / 1e3:.1f} us")
    oth = [k for k in ks if k[0] >= t0 and k[1] <= t1 and "terra_render_kernel" not in k[2]]
    print(f"other kernels in the window: {len(oth)}, total {sum(k[1] - k[0] for k in oth) / 1e6:.2f} ms")


if len(sys.argv) > 2 and sys.argv[1] == "--analyze":
    analyze(sys.argv[2]); sys.exit(0)

import torch  # noqa: F401,E402
import ctypes as C, time, threading  # noqa: E402
from terra_amd import api, runtime, scenes  # noqa: E402
lib = runtime.load()
d = scenes.cornell_box(1920, 1080, 512)
scene = scenes.build_scene(lib, d, counters=False); cam = scenes.camera_of(d)
fb = api.Framebuffer(lib, d.width, d.height)
runtime.check(lib.set_sample_split(scene, int(os.environ.get('TILES_SPLIT', '0'))))
tiles = [(x, y, min(128, d.width - x), min(128, d.height - y)) for y in range(0, d.height, 128) for x in range(0, d.width, 128)]


from concurrent.futures import ThreadPoolExecutor  # noqa: E402
pool = ThreadPoolExecutor(max_workers=8) if os.environ.get("TILES_POOL", "1") != "0" else None      # persistent workers (the client's job system) or fresh threads per frame


def frame():
    def worker(k):
        for t in tiles[k::8]:
            lib.render(C.byref(cam), scene, C.byref(fb.fb), *t)
    t = time.perf_counter()
    if pool:
        list(pool.map(worker, range(8)))
    else:
        ths = [threading.Thread(target=worker, args=(k,)) for k in range(8)]
        [th.start() for th in ths]; [th.join() for th in ths]
    return time.perf_counter() - t


frame()
if os.environ.get("TILES_REPEAT"):
    print("frames wall ms:", [round(frame() * 1e3, 2) for _ in range(int(os.environ["TILES_REPEAT"]))])
print("frame wall ms:", round(frame() * 1e3, 2), "error:", runtime.last_error())
