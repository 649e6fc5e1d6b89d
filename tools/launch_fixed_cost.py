"""What a device-resident render launch costs beside its pixels (dev tool, GPU box): rectangles of growing size at the headline's 512 spp / split 32, HIP-event time per
launch, least-squares fit T = a + b * pixels. `a` is what every shard of an N-GPU frame and every tile call pays once.
    python3 tools/launch_fixed_cost.py [spp] [split]"""
import torch, ctypes as C, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from terra_amd import api, runtime, scenes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 512
split = int(sys.argv[2]) if len(sys.argv) > 2 else 32
L = runtime.load()
d = scenes.cornell_box(1920, 1080, spp)
scene = scenes.build_scene(L, d, counters=False); cam = scenes.camera_of(d)
L.set_sample_split(scene, split)
fb = runtime.DeviceFramebuffer(d.width, d.height)
rows = []
for (w, h) in ((64, 64), (128, 128), (256, 256), (512, 256), (512, 512), (1024, 512), (1920, 540), (1920, 1080)):
    x0, y0 = (d.width - w) // 2 // 64 * 64, (d.height - h) // 2 // 64 * 64          # centred: the expensive part of the image
    for _ in range(2): runtime.render_device(L, cam, scene, fb, (x0, y0, w, h))
    torch.cuda.synchronize()
    n = 5; e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): runtime.render_device(L, cam, scene, fb, (x0, y0, w, h))
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    rows.append((w * h, ms)); print(f"{w:5d} x {h:4d}  {ms:8.3f} ms  {w*h*spp/ms/1e6:9.1f} Gsamples/s" if False else f"{w:5d} x {h:4d}  {ms:8.3f} ms  {w*h*spp/ms/1e3:9.1f} Msamples/s", flush=True)
A = np.array([[1.0, p] for p, _ in rows]); y = np.array([m for _, m in rows])
a, b = np.linalg.lstsq(A, y, rcond=None)[0]
print(f"fit: {a:.3f} ms + {b*1e6:.4f} ms per Mpixel   (spp {spp}, split {split}); error {runtime.last_error()!r}")
