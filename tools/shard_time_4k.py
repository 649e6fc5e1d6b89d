"""Config 5 (hall, 3840x2160, 4096 spp) as the 8-GPU job sees it: time of ONE rank's share (tiles t % 8 == rank) on one GPU, per sample split.
    python tools/shard_time_4k.py [--spp 4096] [--ranks 0,3]"""
import torch  # first: libterra_amd.so must bind to the HIP runtime torch loaded
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from terra_amd import runtime, scenes

ap = argparse.ArgumentParser(); ap.add_argument("--spp", type=int, default=4096); ap.add_argument("--ranks", default="0,3"); ap.add_argument("--splits", default="1,2,4,8")
a = ap.parse_args()
lib = runtime.load()
d = scenes.sponza_hall(3840, 2160, a.spp, bounces=8)
scene = scenes.build_scene(lib, d); cam = scenes.camera_of(d)
fb = runtime.DeviceFramebuffer(d.width, d.height)
for split in [int(x) for x in a.splits.split(",")]:
    runtime.check(lib.set_sample_split(scene, split))
    for r in [int(x) for x in a.ranks.split(",")]:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); runtime.render_device_sharded(lib, cam, scene, fb, 64, r, 8); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        print(f"split {split} rank {r}/8: {ms:9.1f} ms   {d.width * d.height * a.spp / 8 / ms / 1e3:8.1f} Msamples/s per GPU   x8 = {d.width * d.height * a.spp / ms / 1e3:8.1f}", flush=True)
