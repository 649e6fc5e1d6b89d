#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's config.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (config.workload): BASELINE.json configs[1] -- the Cornell box at
1920x1080, 512 spp, max 8 bounces (Simple integrator, random sampler, jitter 0.5,
tonemap None) -- rendered through the product's
device-resident entry point terra_amd_render_device[_sharded] (the framebuffer
is already in HBM when the timed region starts; terra_render()'s PCIe-inclusive
rate is reported in DESIGN.md, never here).

A step = one pass of the hot path over the whole frame: every pixel receives
spp more samples. With N ranks (one process per GPU) the frame's 64x64 tiles are
dealt round-robin to the ranks (t % N == rank), each rank renders its tiles on
its own scene replica, and ONE gather (RCCL over xGMI) moves the packed tiles to
rank 0, which unpacks them into the full frame. Total work is fixed as N grows:
"scaling": "strong". value = frame samples * K / max-over-ranks wall time.
Every rank count renders with the same sample split (terra_amd_set_sample_split, default 8 lanes per pixel:
the frame of 8 successive 64-spp calls), so the image does not depend on N and a 1/8 share of the frame
still fills a GPU.

Also on the JSON line (rank 0):
  roofline     -- the render kernel's ALGORITHMIC bytes per launch (device work
                  counters x SURVEY.md 8d's per-unit sizes) / its average launch
                  duration measured with HIP events on the launch stream, against
                  the 8 TB/s HBM peak; traffic = measured HBM bytes per launch
                  from the committed rocprofv3 PMC passes (profiles/), or null.
  cpu_baseline -- the reference's own CPU renderer (oracle/_ref, prebuilt from its unmodified
                  sources; kind "reference") on all usable host cores over a bounded crop of the
                  same workload, with the oracle's rate beside it (port_value); the oracle alone
                  (kind "port") when the prebuilt reference library is absent. N=1 only.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

import torch  # before the library: it must bind to the HIP runtime torch loads

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

from terra_amd import api, runtime, scenes  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
TILE = 64


def algorithmic_bytes(st: dict) -> float:
    """64 B per node popped, 36 B per triangle test, 36+60 B per hit,
    12 B per attribute fetched, 44 B per pixel per call (reference struct sizes, not the padded device ones)."""
    return 64.0 * st["nodes"] + 36.0 * st["tri_tests"] + 96.0 * st["hits"] + 12.0 * st["attr_fetches"] + 44.0 * st["pixels"]


def workload(name: str, spp_override):
    if name == "cornell_1080p_512spp":
        d = scenes.cornell_box(1920, 1080, 512, bounces=8, integrator=api.kTerraIntegratorSimple)
    elif name == "cornell_256_4spp":
        d = scenes.cornell_box(256, 256, 4, bounces=8, integrator=api.kTerraIntegratorSimple)
    elif name == "cornell_1080p_512spp_direct":
        d = scenes.cornell_box(1920, 1080, 512, bounces=8, integrator=api.kTerraIntegratorDirect)
    elif name == "cornell_phong_1080p_512spp":   # the Cornell box with Phong boxes (what OBJ/MTL scenes with Ks map to)
        d = scenes.cornell_phong(1920, 1080, 512, bounces=8, integrator=api.kTerraIntegratorSimple)
    elif name == "hall_1080p_256spp":        # BASELINE.json configs[2]: ~100k triangles, deep reference-tree traversal
        d = scenes.sponza_hall(1920, 1080, 256, bounces=8, integrator=api.kTerraIntegratorSimple)
    elif name == "hall_2160p_4096spp":       # BASELINE.json configs[4]: the 100k scene at 3840x2160, 4096 spp (meant for 8 GPUs)
        d = scenes.sponza_hall(3840, 2160, 4096, bounces=8, integrator=api.kTerraIntegratorSimple)
    elif name == "spheres_1080p_1024spp":    # BASELINE.json configs[3]: glass + GGX spheres (this repo's presets; no reference behaviour)
        d = scenes.cornell_spheres(1920, 1080, 1024, bounces=8, integrator=api.kTerraIntegratorSimple)
    else:
        raise SystemExit(f"unknown workload {name}")
    if spp_override:
        d.spp = spp_override
    return d


def usable_cores() -> int:
    """host cores this process may actually use: affinity mask capped by the cgroup CPU quota"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = Path(path).read_text().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                q = int(txt[0]); p = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
                if q > 0:
                    n = min(n, max(1, int(q / p + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def _time_cpu(lib, prefix, d, cores, seconds_budget, rows_total=None):
    """one CPU renderer (same C entry-point shape for the compiled reference and the oracle) over a stratified sample of the
    frame at full spp: up to 8 full-width bands evenly spaced over the height, so that the sample's mix of cheap (background)
    and expensive (inside the box) pixels is the frame's. Returns (Msamples/s, description, rows used)."""
    f = lib.fn(prefix + "render_pixels_mt", None, [C.POINTER(api.TerraCamera), C.c_void_p, C.POINTER(api.TerraFramebuffer)] + [C.c_size_t] * 4 + [C.c_uint64, C.c_void_p, C.c_int])
    scene = scenes.build_scene(lib, d)
    cam = scenes.camera_of(d)
    fb = api.Framebuffer(lib, d.width, d.height)
    BANDS = max(1, min(8, cores // 2))
    per_band_threads = max(1, cores // BANDS)

    def run(rows):
        # the bands render concurrently (ctypes drops the GIL), each on its share of the cores, so short bands still use every core
        import threading
        per = max(1, rows // BANDS); fb.clear()
        def one(b):
            y0 = min(d.height - per, max(0, int((b + 0.5) * d.height / BANDS) - per // 2))
            f(C.byref(cam), scene, C.byref(fb.fb), 0, y0, d.width, per, scenes.FRAME_SEED, None, per_band_threads)
        ths = [threading.Thread(target=one, args=(b,)) for b in range(BANDS)]
        t = time.perf_counter(); [th.start() for th in ths]; [th.join() for th in ths]
        return per * BANDS, time.perf_counter() - t

    if rows_total is None:          # size the sample for ~seconds_budget: a thin pass, then one re-sizing pass if it came out short
        rows, dt = run(BANDS * 4 * per_band_threads)
        if dt < 0.6 * seconds_budget:           # (a heavy workload can exhaust the budget with the thin pass alone: then that is the sample)
            rows_total = int(min(d.height, max(BANDS, rows * seconds_budget / max(dt, 1e-3))))
            rows, dt = run(rows_total)
        if dt < 0.6 * seconds_budget and rows < d.height:
            rows_total = int(min(d.height, rows * seconds_budget / max(dt, 1e-3)))
            rows, dt = run(rows_total)
    else:
        rows, dt = run(rows_total)
    val = d.width * rows * d.spp / dt / 1e6
    fb.destroy(); lib.scene_destroy(scene)
    return val, f"{BANDS} full-width bands of {rows // BANDS} rows evenly spaced over the {d.width}x{d.height} frame ({rows} rows), full {d.spp} spp, {dt:.1f} s", rows


def cpu_baseline(d: scenes.SceneDesc, seconds_budget: float = 12.0):
    """The reference's own CPU renderer (oracle/_ref/libterra_ref.so: its unmodified sources compiled in the build
    container with per-pixel pinned entropy; kind "reference") when that prebuilt library travelled with the tree, and the
    oracle (bit-exact CPU restatement; kind "port") -- both on every usable host core, each over a bounded crop."""
    import subprocess
    cores = usable_cores()
    subprocess.run(["make", "-C", str(ROOT / "oracle")], check=True, capture_output=True)
    orc = api.TerraLib(ROOT / "oracle" / "liboracle.so", "orc_")
    port, port_sample, rows = _time_cpu(orc, "orc_", d, cores, seconds_budget)
    ref_so = ROOT / "oracle" / "_ref" / "libterra_ref.so"
    if ref_so.exists() and all(o.material.kind in ("diffuse", "phong") for o in d.objects):    # the reference has no GGX/glass preset
        val, sample, _ = _time_cpu(api.TerraLib(ref_so, "terra_"), "ref_", d, cores, seconds_budget, rows_total=rows)     # the same rows as the oracle's run
        return {"value": round(val, 3), "unit": "Msamples/s", "cores": cores, "kind": "reference",
                "sample": sample + f", oracle/_ref/libterra_ref.so (the reference's sources, gcc -O2, one terra_render call per pixel) on {cores} threads",
                "port_value": round(port, 3), "port_sample": port_sample + f", oracle/liboracle.so on {cores} threads"}
    return {"value": round(port, 3), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": port_sample + f", oracle/liboracle.so on {cores} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cornell_1080p_512spp")
    ap.add_argument("--integrator", default="", choices=["", "simple", "direct", "mis"], help="override the workload's integrator (the result is then NOT the headline config)")
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel (the result is then NOT the headline config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, the real path) or gloo (rehearsal of N ranks on fewer GPUs: the gather goes through host memory)")
    ap.add_argument("--tree", default="reference", choices=["reference", "fast"], help="reference = the reference's own tree and traversal order (parity mode, the headline); fast = terra_amd_set_tree_mode(1)")
    ap.add_argument("--sample-split", type=int, default=8, help="terra_amd_set_sample_split: lanes per pixel (the frame equals that of this many successive calls of spp/split samples); the same for every N so the image does not depend on N")
    ap.add_argument("--check", action="store_true", help="after timing: one sharded+gathered pass on a cleared frame must equal an unsharded pass bit for bit (rank 0)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    import torch.distributed as dist
    # the multi-rank code path; TERRA_BENCH_DIST1=1 takes it with ONE rank too (shard, pack, RCCL gather, unpack): a self-test of
    # those calls on a 1-GPU box, never a benchmark
    dist_on = world > 1 or os.environ.get("TERRA_BENCH_DIST1") == "1"
    ngpu = torch.cuda.device_count()
    if ngpu < 1:
        raise SystemExit("bench.py needs an MI355X")
    dev_index = (local_rank % ngpu) if world > 1 else 0        # ranks > GPUs only happens in a gloo rehearsal
    torch.cuda.set_device(dev_index)
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world)
    dev = torch.device("cuda", dev_index)
    via_host = dist_on and args.dist_backend != "nccl"

    lib = runtime.load()
    if lib.device_count() <= 0:
        raise SystemExit("bench.py needs an MI355X: " + runtime.last_error())
    runtime.check(lib.set_device(dev.index), "terra_amd_set_device")

    d = workload(args.workload, args.spp)
    if args.integrator:
        d.integrator = {"simple": api.kTerraIntegratorSimple, "direct": api.kTerraIntegratorDirect, "mis": api.kTerraIntegratorDirectMis}[args.integrator]
    scene = scenes.build_scene(lib, d, tree_mode=1 if args.tree == "fast" else 0)
    if runtime.last_error():
        raise SystemExit("scene commit failed: " + runtime.last_error())
    runtime.check(lib.set_sample_split(scene, args.sample_split), "terra_amd_set_sample_split")
    cam = scenes.camera_of(d)
    fb = runtime.DeviceFramebuffer(d.width, d.height, device=dev)
    n_packed = runtime.packed_floats_per_rank(d.width, d.height, TILE, world)
    packed = torch.zeros(n_packed, dtype=torch.float32, device=dev) if dist_on else None
    gather_bufs = [torch.zeros(n_packed, dtype=torch.float32, device=dev) for _ in range(world)] if (dist_on and rank == 0) else None
    stream = torch.cuda.current_stream(dev).cuda_stream
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

    def step(i=None):
        if i is not None:
            ev[i][0].record()
        if not dist_on:
            runtime.check(lib.render_device(C.byref(cam), scene, fb.pixels.data_ptr(), fb.results.data_ptr(), d.width, d.height, 0, 0, d.width, d.height, None, stream), "render")
        else:
            runtime.check(lib.render_device_sharded(C.byref(cam), scene, fb.pixels.data_ptr(), fb.results.data_ptr(), d.width, d.height, 0, 0, d.width, d.height, TILE, rank, world, None, stream), "render")
        if i is not None:
            ev[i][1].record()
        if dist_on:
            runtime.check(lib.pack_tiles(fb.pixels.data_ptr(), fb.results.data_ptr(), d.width, d.height, 0, 0, d.width, d.height, TILE, rank, world, packed.data_ptr(), stream), "pack")
            if via_host:
                torch.cuda.synchronize(dev)
                hb = [torch.zeros(n_packed, dtype=torch.float32) for _ in range(world)] if rank == 0 else None
                dist.gather(packed.cpu(), gather_list=hb, dst=0)
                if rank == 0:
                    for src in range(1, world):
                        gather_bufs[src].copy_(hb[src])
            else:
                dist.gather(packed, gather_list=gather_bufs, dst=0)
            if rank == 0:
                for src in range(1, world):
                    runtime.check(lib.unpack_tiles(fb.pixels.data_ptr(), fb.results.data_ptr(), d.width, d.height, 0, 0, d.width, d.height, TILE, src, world, gather_bufs[src].data_ptr(), stream), "unpack")

    def fence():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    runtime.check(lib.reset_stats(scene))
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    fence()
    elapsed = time.perf_counter() - t0
    if dist_on:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    check = None
    if args.check:
        fb.clear(); fence(); step(); fence()
        if rank == 0:
            ref_fb = runtime.DeviceFramebuffer(d.width, d.height, device=dev)
            runtime.check(lib.render_device(C.byref(cam), scene, ref_fb.pixels.data_ptr(), ref_fb.results.data_ptr(), d.width, d.height, 0, 0, d.width, d.height, None, stream), "render")
            torch.cuda.synchronize(dev)
            check = bool(torch.equal(ref_fb.pixels.view(torch.int32), fb.pixels.view(torch.int32)) and torch.equal(ref_fb.results, fb.results))
    st = runtime.Stats(); runtime.check(lib.get_stats(scene, C.byref(st))); st = st.as_dict()
    kernel_ms = sum(a.elapsed_time(b) for a, b in ev) / args.steps
    frame_samples = d.width * d.height * d.spp
    value = frame_samples * args.steps / elapsed / 1e6

    if rank == 0:
        alg = algorithmic_bytes(st) / max(1, st["launches"])
        achieved = alg / (kernel_ms * 1e-3) / 1e9
        traffic = None
        tf = ROOT / "profiles" / "roofline_traffic.json"
        if tf.exists() and world == 1 and not args.spp:
            rec = json.loads(tf.read_text()).get(args.workload + ("" if args.tree == "reference" else ":" + args.tree), {})
            if rec.get("sample_split", 1) == args.sample_split and not args.integrator:      # the PMC passes were taken on this configuration
                traffic = rec.get("hbm_bytes_per_launch")
        out = {
            "metric": "Msamples/s", "value": round(value, 2), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "scene": d.name, "width": d.width, "height": d.height, "spp": d.spp, "bounces": d.bounces,
                       "integrator": {0: "simple", 1: "direct", 2: "mis"}.get(d.integrator, str(d.integrator)),
                       "triangles": d.triangle_count, "tree": args.tree, "tile": TILE, "sample_split": args.sample_split, "parallelism": f"tiles%{world}" if world > 1 else "single"},
            "mrays_per_s": round(st["rays"] / max(1, st["launches"]) * (1 if world == 1 else world) / (kernel_ms * 1e-3) / 1e6, 1),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "kernel": "terra_render_kernel", "kernel_ms": round(kernel_ms, 3),
                         "algorithmic_bytes_per_launch": int(alg), "rank0_only": world > 1,
                         "note": "achieved = algorithmic bytes (SURVEY 8d) / kernel time; it can exceed the HBM peak because a scene that fits is served from LDS (or L2 / Infinity Cache): `traffic` is what HBM actually moved"},
            "counters_per_launch": {k: v // max(1, st["launches"]) for k, v in st.items() if k != "launches"},
        }
        if check is not None:
            out["sharded_equals_unsharded"] = check
        if dist_on:
            out["dist_backend"] = args.dist_backend
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(d)
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
