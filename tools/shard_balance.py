"""Load balance of the tile deal (t % world == rank) measured on ONE GPU: each rank's share of the
headline frame is rendered by itself and timed with HIP events; max/mean over ranks bounds the
strong-scaling efficiency the shard rule allows (the gather comes on top).
    python tools/shard_balance.py [--spp 64] [--world 8]"""
import torch  # first: libterra_amd.so must bind to the HIP runtime torch loaded
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from terra_amd import api, runtime, scenes

ap = argparse.ArgumentParser(); ap.add_argument("--spp", type=int, default=64); ap.add_argument("--world", type=int, default=8)
ap.add_argument("--workload", default="cornell"); ap.add_argument("--split", type=int, default=1)
a = ap.parse_args()
lib = runtime.load()
d = scenes.cornell_box(1920, 1080, a.spp, bounces=8) if a.workload == "cornell" else scenes.sponza_hall(1920, 1080, a.spp, bounces=8)
scene = scenes.build_scene(lib, d, counters=False); cam = scenes.camera_of(d)      # library defaults: automatic traversal, no work counters
runtime.check(lib.set_sample_split(scene, a.split))
fb = runtime.DeviceFramebuffer(d.width, d.height)
def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize(); e0.record(); fn(); e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)
whole = timed(lambda: runtime.render_device(lib, cam, scene, fb))
print(f"split {a.split}: whole frame {whole:.2f} ms")
for tile in (32, 64, 128):
    for world in sorted({2, 4, a.world}):
        ts = [timed(lambda r=r: runtime.render_device_sharded(lib, cam, scene, fb, tile, r, world)) for r in range(world)]
        print(f"tile {tile:4d} world {world}: max {max(ts):7.2f} mean {sum(ts) / world:7.2f} ms  max/mean {max(ts) * world / sum(ts):.3f}  ideal-speedup-bound {whole / max(ts):.2f}x")
