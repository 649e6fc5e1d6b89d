"""Sample split x share of the frame (dev tool, GPU box): the slowest of `world` shards (64-pixel tiles, t % world == rank) of the Cornell 1080p 512 spp frame for every sample
split, HIP-event ms. What the automatic split (terra_amd_set_sample_split(scene, 0)) should choose for a launch of a given size.
    python3 tools/split_matrix.py [integrator]"""
import torch, sys, os
sys.path.insert(0, os.getcwd())
from terra_amd import api, runtime, scenes
integ = {"simple": api.kTerraIntegratorSimple, "direct": api.kTerraIntegratorDirect}[sys.argv[1] if len(sys.argv) > 1 else "simple"]
L = runtime.load()
d = scenes.cornell_box(1920, 1080, 512, bounces=8, integrator=integ)
scene = scenes.build_scene(L, d, counters=False); cam = scenes.camera_of(d)
fb = runtime.DeviceFramebuffer(d.width, d.height)
def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize(); e0.record(); fn(); fn(); e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / 2
print("split  " + "  ".join(f"world {w:2d} (x{w})" for w in (1, 2, 4, 8, 16)))
for split in (2, 4, 8, 16, 32, 64):
    runtime.check(L.set_sample_split(scene, split))
    row = []
    for world in (1, 2, 4, 8, 16):
        ranks = range(world) if world <= 8 else range(0, world, 2)
        t = max(timed(lambda r=r: runtime.render_device_sharded(L, cam, scene, fb, 64, r, world)) for r in ranks)
        row.append(f"{t:7.2f} ({t * world:6.1f})")
    print(f"{split:5d}  " + "  ".join(row), flush=True)
print("error:", repr(runtime.last_error()))
