"""Kernel timeline of sharded launches (dev tool): rank 0's share of the headline frame at world 1 / 2 / 4 / 8 / 16, three launches each; run under
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/shard_trace -- python3 tools/shard_trace.py
and read with tools/shard_trace.py --read gpurun_out/shard_trace"""
import sys, os, glob, csv, collections
if len(sys.argv) > 2 and sys.argv[1] == "--read":
    f = glob.glob(sys.argv[2] + "/*/*kernel_trace.csv")[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    prev_end = None
    for r in rows:
        name = r["Kernel_Name"].split("(")[0][:40]; s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - prev_end) / 1e3 if prev_end else 0.0
        print(f"{name:42s} grid {r.get('Grid_Size', '?'):>9s} dur {(e - s) / 1e3:10.1f} us  gap before {gap:8.1f} us")
        prev_end = e
    sys.exit(0)
import torch, ctypes as C
sys.path.insert(0, os.getcwd())
from terra_amd import api, runtime, scenes
L = runtime.load()
d = scenes.cornell_box(1920, 1080, 512)
scene = scenes.build_scene(L, d, counters=False); cam = scenes.camera_of(d)
L.set_sample_split(scene, 32)
fb = runtime.DeviceFramebuffer(d.width, d.height)
for world in (1, 2, 4, 8, 16):
    for _ in range(3):
        runtime.render_device_sharded(L, cam, scene, fb, 64, 0, world)
    torch.cuda.synchronize()
