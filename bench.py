#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's config.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

`python bench.py --gpus N` with N > 1 from a bare shell starts the N ranks itself: before anything touches the GPU it
spawns `python -m torch.distributed.run ...` as a CHILD process, relays the child's JSON line and exits with its code.

Workload (config.workload): BASELINE.json configs[1] -- the Cornell box at 1920x1080, 512 spp, max 8 bounces (Simple
integrator, random sampler, jitter 0.5, tonemap None) -- rendered through the product's
device-resident entry point terra_amd_render_device[_sharded] (the framebuffer is already in HBM when the timed region
starts; terra_render()'s PCIe-inclusive rate is reported in DESIGN.md, never here).

A step = one pass of the hot path over the whole frame: every pixel receives spp more samples. With N ranks (one process
per GPU) the frame's 64x64 tiles are dealt round-robin to the ranks (t % N == rank), each rank renders its tiles on its own
scene replica, and ONE gather (RCCL over xGMI; terra_amd.runtime.gather_frame, the function the gloo test covers) moves the
packed tiles to rank 0, which unpacks them into the full frame. Pack, gather and unpack run on a second stream: the next
step's render (disjoint tiles) starts as soon as the pack has read the rank's own tiles. Total work is fixed as N grows:
"scaling": "strong". value = frame samples * K / max-over-ranks wall time. Every rank count renders with the library's AUTOMATIC sample split
for a launch of its size (terra_amd_auto_sample_split: 8 / 16 / 32 / 32 lanes per pixel at 1 / 2 / 4 / 8 ranks on the headline frame -- the frame of that many
successive calls of spp / split samples; --sample-split S fixes it for every N), so the image does not
depend on N and a 1/8 share of the frame still fills a GPU.

Also on the JSON line (rank 0):
  roofline     -- against the resource that binds the render kernel (DESIGN.md "Roofline"):
                  bound "valu" for scenes staged in LDS: achieved = VALU wave-instructions per second (SQ_INSTS_VALU of the
                  committed PMC summary profiles/rNN_pmc.json / the kernel time measured live with HIP events), peak = 1024
                  SIMDs x 2.4 GHz / 2 cycles per wave64 instruction; lane_util and the LDS bank-conflict share beside it;
                  bound "l2_fabric" for scenes read from global memory: achieved = bytes the L2 moved on its fabric side
                  (FETCH_SIZE/WRITE_SIZE PMC passes, Infinity Cache hits included) per second, peak = the 8 TB/s HBM figure.
                  algorithmic_* = SURVEY.md 8d's per-unit bytes x the device work counters, kept as its own field.
  cpu_baseline -- the reference's own CPU renderer (oracle/_ref, prebuilt from its unmodified sources; kind "reference")
                  on all usable host cores over a bounded sample of the same workload, with the oracle's rate beside it
                  (port_value); the oracle alone (kind "port") when the prebuilt reference library is absent. N=1 only.
  host_api     -- (N=1) the same frame through the drop-in boundary, terra_render() on a host framebuffer, PCIe included: one
                  full-frame call per step, and the reference client's 128-pixel tile loop from 8 threads. Never `value`.
  phases       -- (N>1) per rank: render / pack / gather / unpack milliseconds per step (HIP events on the two streams), the
                  slowest rank, and sharded_equals_unsharded (one extra low-spp step, checked bit for bit on rank 0)
  workloads    -- (N=1, default invocation) the other BASELINE.json configurations measured the same way in the same run,
                  each with its own value / ms_per_step / roofline / cpu_baseline. The headline stays configs[1].
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # before the HIP runtime starts (terra_amd/runtime.py says why): the host_api block's 8 caller threads each own a stream

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_GINST = 1024 * 2.4 / 2      # 256 CUs x 4 SIMD-32, a wave64 VALU instruction issues over 2 cycles at 2.4 GHz (MI355X_MICROARCH.md "Execution model")
TA_PEAK_GCYC = 256 * 2.4              # one texture addresser per CU, 2.4 GHz: G addresser-cycles per second
L2_GATHER_GBS = 18800.0               # MI355X_MICROARCH.md "Indexed rows": rows gathered from the XCDs' L2, 16.8-18.8 TB/s chip-wide (the upper figure: the replica kernel moves 17.5)
TILE = 64
# Sample split of the timed launches (terra_amd_set_sample_split): the frame equals that of this many successive calls of spp/split samples. 0 = the library's automatic choice
# for a launch of this size (terra_amd_auto_sample_split; measure() resolves it to a number so that the sharded and the unsharded frames of a run share it and the line reports
# it): with the job order of round 4 a launch of an LDS-resident scene wants ~50 jobs per resident lane, the others ~200 -- slowest of 1 / 2 / 4 / 8 shards of the Cornell frame
# 50.2 / 25.6 / 13.5 / 7.0 ms at the automatic 8 / 16 / 32 / 32 against 53.5 / 26.9 / 13.5 / 7.0 at 32 everywhere (profiles/r04_measurements/split_matrix.log)
DEFAULT_SPLIT = 0
TREE_MODES = {"auto": 2, "reference": 0, "fast": 1}
INTEGRATORS = {"simple": 0, "direct": 1, "mis": 2}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cornell_1080p_512spp")
    ap.add_argument("--integrator", default="", choices=["", "simple", "direct", "mis"], help="override the workload's integrator (the result is then NOT the headline config)")
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel (the result is then NOT the headline config)")
    ap.add_argument("--bounces", type=int, default=-1, help="override the maximum number of bounces (the result is then NOT the headline config; 0 = camera rays only)")
    ap.add_argument("--job-order", type=int, default=-1, choices=[-1, 0, 1], help="terra_amd_set_job_order: -1 the library's default (on), 0 / 1 to A/B it (the frame does not change)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-api", action="store_true", help="skip the `host_api` block (the same frame through terra_render() on a host framebuffer)")
    ap.add_argument("--no-workloads", action="store_true", help="skip the `workloads` block (the other configurations)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, the real path) or gloo (rehearsal of N ranks on fewer GPUs: the gather goes through host memory)")
    ap.add_argument("--tree", default="auto", choices=list(TREE_MODES), help="terra_amd_set_tree_mode: auto (2, the library default: leaf-box cull / fast tree when the scene passes the numeric containment check), reference (0: the reference's tree, every traversal decision reproduced), fast (1)")
    ap.add_argument("--sample-split", type=int, default=DEFAULT_SPLIT, help="terra_amd_set_sample_split: lanes per pixel (the frame equals that of this many successive calls of spp/split samples); 0 (default) = the library's automatic choice for the launch each rank makes (terra_amd_auto_sample_split), a number = the same for every N")
    ap.add_argument("--check", action="store_true", help="after timing: one low-spp sharded+gathered pass into a fresh frame must equal an unsharded pass bit for bit (rank 0); on by default when N > 1")
    ap.add_argument("--no-check", action="store_true", help="N > 1: skip that extra pass")
    ap.add_argument("--launch-timeout", type=float, default=900.0, help="seconds the self-started N-rank child may run before its process group is killed (a hung rendezvous must not hang the parent; well below the driver's own 1500 s limit, so that the ranks are gone before the driver kills this parent)")
    ap.add_argument("--master-port", type=int, default=0, help="rendezvous port of the self-started ranks (0 = pick a free one)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------------------------
# starting the ranks ourselves (no GPU call may precede this: the ranks are CHILD processes, nothing is exec'ed)
# ------------------------------------------------------------------------------------------------------------------

def launcher_command(args, argv, port):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
            str(Path(__file__).resolve())] + list(argv)


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _die_with_parent():
    """preexec of the launcher child: SIGKILL when this process dies without having stopped it (the handlers below cannot run on SIGKILL)"""
    try:
        import ctypes
        ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, 9, 0, 0, 0)          # PR_SET_PDEATHSIG, SIGKILL
    except OSError:
        pass


def self_launch(args, argv) -> int:
    """bench.py --gpus N (N > 1) without a launcher: run the N ranks as a child torch.distributed.run in its own process group, so that the WHOLE group can be
    killed -- after --launch-timeout (a hung rendezvous), or when this parent is told to stop: SIGTERM / SIGINT / SIGHUP aimed at the parent are forwarded as a
    SIGKILL of the group (the ranks live in another session and would otherwise keep the GPUs) -- pass the ranks' stderr through live, relay rank 0's JSON line"""
    import signal
    port = args.master_port or free_port()
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0"); env.setdefault("OMP_NUM_THREADS", "4")
    p = subprocess.Popen(launcher_command(args, argv, port), env=env, stdout=subprocess.PIPE, stderr=None, text=True, start_new_session=True, preexec_fn=_die_with_parent)

    def kill_group():
        try:
            os.killpg(p.pid, signal.SIGKILL)            # the exact group this call started
        except ProcessLookupError:
            pass

    def on_signal(signum, _frame):
        kill_group()
        sys.stderr.write(f"bench.py: signal {signum}: the {args.gpus}-rank child's process group was killed\n")
        os._exit(128 + signum)
    old = {sg: signal.signal(sg, on_signal) for sg in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP)}
    try:
        try:
            stdout, _ = p.communicate(timeout=args.launch_timeout)
        except subprocess.TimeoutExpired:
            kill_group()
            stdout, _ = p.communicate()
            sys.stderr.write((stdout or "")[-4000:] + f"\nbench.py: the {args.gpus}-rank child did not finish within {args.launch_timeout:.0f} s and was killed\n")
            return 124
    finally:
        for sg, h in old.items():
            signal.signal(sg, h)
    line = None
    for ln in (stdout or "").splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    if p.returncode != 0 or line is None:
        sys.stderr.write((stdout or "")[-4000:] + "\n")
        sys.stderr.write(f"bench.py: the {args.gpus}-rank child exited with code {p.returncode}" + ("" if line else " and printed no result line") + "\n")
        return p.returncode or 1
    print(line, flush=True)
    return 0


# ------------------------------------------------------------------------------------------------------------------
# workloads, bytes, CPU baseline
# ------------------------------------------------------------------------------------------------------------------

def algorithmic_bytes(st: dict) -> float:
    """64 B per node popped, 36 B per triangle test, 36+60 B per hit,
    12 B per attribute fetched, 44 B per pixel per call (reference struct sizes, not the padded device ones)."""
    return 64.0 * st["nodes"] + 36.0 * st["tri_tests"] + 96.0 * st["hits"] + 12.0 * st["attr_fetches"] + 44.0 * st["pixels"]


def workload(name: str, spp_override=0):
    from terra_amd import api, scenes
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
    elif name == "hall_1080p_64spp_direct":  # the same hall with the reference client's default integrator (satellite/include/Config.hpp), fewer samples to keep the run short
        d = scenes.sponza_hall(1920, 1080, 64, bounces=8, integrator=api.kTerraIntegratorDirect)
    elif name == "hall_2160p_4096spp":       # BASELINE.json configs[4]: the 100k scene at 3840x2160, 4096 spp (meant for 8 GPUs)
        d = scenes.sponza_hall(3840, 2160, 4096, bounces=8, integrator=api.kTerraIntegratorSimple)
    elif name == "hall_x100_1080p_64spp":    # the hall with every coordinate (scene and camera) x 100: outside the +-13-unit range of the containment proof, the
        # automatic mode keeps the fast tree and replays the reference's reachability for the closest hit
        import numpy as np
        d = scenes.sponza_hall(1920, 1080, 64, bounces=8, integrator=api.kTerraIntegratorSimple)
        for o in d.objects:
            o.triangles = (np.asarray(o.triangles, np.float32) * np.float32(100.0)).astype(np.float32)
        d.camera_position = tuple(float(np.float32(c) * np.float32(100.0)) for c in d.camera_position)
        d.name = "sponza_hall_x100"
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


def _time_cpu(lib, prefix, d, cores, seconds_budget, plan=None):
    """one CPU renderer (same C entry-point shape for the compiled reference and the oracle) over a stratified sample of the
    frame: up to 8 full-width bands evenly spaced over the height (so the sample's mix of cheap and expensive pixels is the
    frame's), at the workload's full spp when the budget allows and at a reduced spp (stated) for the heavy scenes, where a
    single full-spp row would already exceed it. Returns (Msamples/s, description, plan) -- plan = (rows, spp) for re-use."""
    import threading
    from terra_amd import api, scenes
    f = lib.fn(prefix + "render_pixels_mt", None, [C.POINTER(api.TerraCamera), C.c_void_p, C.POINTER(api.TerraFramebuffer)] + [C.c_size_t] * 4 + [C.c_uint64, C.c_void_p, C.c_int])
    BANDS = max(1, min(8, cores // 2))
    per_band_threads = max(1, cores // BANDS)
    cam = scenes.camera_of(d)
    full_spp = d.spp

    def run(rows, spp):
        dd = scenes.SceneDesc(**{**d.__dict__, "spp": spp})
        scene = scenes.build_scene(lib, dd)
        fb = api.Framebuffer(lib, d.width, d.height)
        per = max(1, rows // BANDS)
        def one(b):     # the bands render concurrently (ctypes drops the GIL), each on its share of the cores
            y0 = min(d.height - per, max(0, int((b + 0.5) * d.height / BANDS) - per // 2))
            f(C.byref(cam), scene, C.byref(fb.fb), 0, y0, d.width, per, scenes.FRAME_SEED, None, per_band_threads)
        ths = [threading.Thread(target=one, args=(b,)) for b in range(BANDS)]
        t = time.perf_counter(); [th.start() for th in ths]; [th.join() for th in ths]
        dt = time.perf_counter() - t
        fb.destroy(); lib.scene_destroy(scene)
        return per * BANDS, dt

    if plan is None:
        rows, dt = run(BANDS, 1)                                      # probe: one row per band at 1 spp
        rate = d.width * rows / max(dt, 1e-4)                          # samples per second (pessimistic: thread start-up included)
        budget = rate * seconds_budget
        if budget >= d.width * BANDS * full_spp:                       # full spp fits: as many rows as the budget buys
            plan = (int(min(d.height, max(BANDS, budget / (d.width * full_spp)))), full_spp)
        else:                                                          # heavy scene: one row per band, as many of the spp as the budget buys
            plan = (BANDS, int(max(1, min(full_spp, budget / (d.width * BANDS)))))
        rows, dt = run(*plan)
        if dt < 0.5 * seconds_budget and (plan[0] < d.height or plan[1] < full_spp):       # the probe under-estimated: one re-sizing pass
            k = seconds_budget / max(dt, 1e-3)
            plan = (plan[0], int(min(full_spp, max(1, plan[1] * k)))) if plan[1] < full_spp else (int(min(d.height, plan[0] * k)), full_spp)
            rows, dt = run(*plan)
    else:
        rows, dt = run(*plan)
    spp = plan[1]
    val = d.width * rows * spp / dt / 1e6
    what = f"full {full_spp} spp" if spp == full_spp else f"{spp} of the {full_spp} spp"
    return val, f"{BANDS} full-width bands of {rows // BANDS} rows evenly spaced over the {d.width}x{d.height} frame ({rows} rows), {what}, {dt:.1f} s", plan


def cpu_baseline(d, seconds_budget: float = 12.0):
    """The reference's own CPU renderer (oracle/_ref/libterra_ref.so: its unmodified sources compiled in the build
    container with per-pixel pinned entropy; kind "reference") when that prebuilt library travelled with the tree, and the
    oracle (bit-exact CPU restatement; kind "port") -- both on every usable host core, each over the same bounded sample."""
    from terra_amd import api
    cores = usable_cores()
    subprocess.run(["make", "-C", str(ROOT / "oracle")], check=True, capture_output=True)
    orc = api.TerraLib(ROOT / "oracle" / "liboracle.so", "orc_")
    port, port_sample, plan = _time_cpu(orc, "orc_", d, cores, seconds_budget)
    ref_so = ROOT / "oracle" / "_ref" / "libterra_ref.so"
    if ref_so.exists() and all(o.material.kind in ("diffuse", "phong") for o in d.objects):    # the reference has no GGX/glass preset
        val, sample, _ = _time_cpu(api.TerraLib(ref_so, "terra_"), "ref_", d, cores, seconds_budget, plan=plan)     # the same sample as the oracle's run
        return {"value": round(val, 4), "unit": "Msamples/s", "cores": cores, "kind": "reference",
                "sample": sample + f", oracle/_ref/libterra_ref.so (the reference's sources, gcc -O2, one terra_render call per pixel) on {cores} threads",
                "port_value": round(port, 4), "port_sample": port_sample + f", oracle/liboracle.so on {cores} threads"}
    return {"value": round(port, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": port_sample + f", oracle/liboracle.so on {cores} threads"}


# ------------------------------------------------------------------------------------------------------------------
# roofline
# ------------------------------------------------------------------------------------------------------------------

def source_digest() -> str:
    """digest of the kernel sources: a PMC record taken on other sources is flagged stale"""
    import hashlib
    h = hashlib.sha256()
    for f in sorted((ROOT / "terra_amd" / "csrc").glob("*")):
        if f.suffix in (".h", ".hip", ".cpp"):
            h.update(f.read_bytes())
    return h.hexdigest()[:16]


def pmc_key(workload_name, tree, integrator, split, spp_override):
    return f"{workload_name}|tree={tree}|integrator={integrator}|split={split}" + (f"|spp={spp_override}" if spp_override else "")


def pmc_file():
    """the newest committed PMC summary (profiles/rNN_pmc.json, written by tools/profile_round.py)"""
    fs = sorted((ROOT / "profiles").glob("r[0-9][0-9]_pmc.json"))
    return fs[-1] if fs else None


# cycles one wave64 VALU instruction occupies a SIMD's issue, by class, at 2.4 GHz: measured per instruction with 4 waves per SIMD on every CU
# (tools/ubench/valu_rates.hip, profiles/r02_measurements/valu_rates.log). The dynamic class counts come from the SQ_INSTS_VALU_* counters (tools/profile_round.py passes
# mix1 / mix2); what they do not name ("other": compares, selects, min / max, permutes, moves, shifts) is priced with the static mix of exactly those opcodes in the
# kernel's code object (rec["other_cycles"], written by tools/profile_round.py), 4.1 when that is absent.
VALU_CLASS_CYCLES = {"SQ_INSTS_VALU_ADD_F32": 2.55, "SQ_INSTS_VALU_MUL_F32": 2.49, "SQ_INSTS_VALU_FMA_F32": 4.0, "SQ_INSTS_VALU_TRANS_F32": 8.1, "SQ_INSTS_VALU_ADD_F64": 4.31,
                     "SQ_INSTS_VALU_MUL_F64": 4.14, "SQ_INSTS_VALU_FMA_F64": 4.14, "SQ_INSTS_VALU_TRANS_F64": 16.2, "SQ_INSTS_VALU_CVT": 4.2, "SQ_INSTS_VALU_INT32": 3.6, "SQ_INSTS_VALU_INT64": 4.4,
                     "SQ_INSTS_VALU_ADD_F16": 4.2, "SQ_INSTS_VALU_MUL_F16": 4.2, "SQ_INSTS_VALU_FMA_F16": 4.2, "SQ_INSTS_VALU_TRANS_F16": 8.1}


# What a MIX of instruction classes can issue, measured: synthetic streams of independent instructions with the kernels' class proportions, 5 waves per SIMD on every CU
# (tools/ubench/valu_rates.hip `mix`, profiles/r04_measurements/valu_mix_rates.log), in G wave64 instructions per second per SIMD:
#   the headline kernel's dynamic mix (add 12 %, mul 15 %, fma 20 %, transcendental 2 %, integer 17 %, compares / selects / min-max / moves 32 %)   0.936  = 2.56 cycles
#   the fast tree's 4-wide node step (v_fma_mix_f32, v_perm_b32, min / max, compares, selects, integer min / max)                                  0.702  = 3.42 cycles
# The sum of the pure-stream costs of the same instructions (the additive model below) says 3.53 cycles for the first mix: classes overlap in issue, pure-stream rates do not add.
# Between and beyond the two measured mixes the ceiling is interpolated by the one thing that separates them -- the share of the 2.3-2.6-cycle instructions (f32 add / mul) in the
# kernel's dynamic mix: (share, cycles per instruction at 5 waves per SIMD) = (0, 3.42: the node-step stream), (0.27, 2.56: the headline stream), (1, 2.26: a pure v_add_f32 stream)
MIX_CEILING_POINTS = [(0.0, 3.42), (0.2702, 2.56), (1.0, 2.26)]


def valu_mix_ceiling(rec, lds_resident=True):
    """The VALU issue ceiling of this kernel's dynamic instruction mix as a fraction of the nominal peak (2 cycles per wave64 instruction): `frac_of_peak` from the measured mixed
    streams (above), interpolated by the kernel's share of f32 add / mul instructions; `additive_model` = total / sum(class count x pure-stream class cycles) x 2 from the same
    counters, kept beside it because it is the model the round-3 notes used -- the kernels issue FASTER than it allows, which is what shows that it is not a ceiling"""
    pmc = rec.get("pmc", {})
    total = pmc.get("SQ_INSTS_VALU") or rec.get("SQ_INSTS_VALU")
    named = {k: pmc[k] for k in VALU_CLASS_CYCLES if k in pmc}
    if not total or len(named) < 6:
        return None
    other = max(0.0, total - sum(named.values()))
    cycles = sum(v * VALU_CLASS_CYCLES[k] for k, v in named.items()) + other * float(rec.get("other_cycles", 4.1))
    share = (pmc.get("SQ_INSTS_VALU_ADD_F32", 0.0) + pmc.get("SQ_INSTS_VALU_MUL_F32", 0.0)) / total
    pts = MIX_CEILING_POINTS
    for (x0, y0), (x1, y1) in zip(pts, pts[1:]):
        if share <= x1:
            break
    cyc = y0 + (y1 - y0) * (min(max(share, x0), x1) - x0) / (x1 - x0)
    return {"frac_of_peak": round(2.0 / cyc, 4), "cycles_per_instr": round(cyc, 3), "add_mul_share": round(share, 4), "source": "profiles/r04_measurements/valu_mix_rates.log",
            "additive_model": {"frac_of_peak": round(2.0 * total / cycles, 4), "mean_cycles_per_instr": round(cycles / total, 3)}, "other_share": round(other / total, 4),
            "classes": {k.replace("SQ_INSTS_VALU_", ""): round(v / total, 4) for k, v in named.items() if v}}


def roofline(key, st, kernel_ms, lds_resident, world, pmc=None):
    """st: device work counters per launch; the PMC record (per launch of the render kernel) comes from the newest profiles/rNN_pmc.json"""
    alg = algorithmic_bytes(st)
    rec = {}
    f = Path(pmc) if pmc else pmc_file()
    if f is not None and f.exists():
        rec = json.loads(f.read_text()).get(key, {})
    t = kernel_ms * 1e-3
    out = {"kernel": "terra_render_kernel", "kernel_ms": round(kernel_ms, 3), "rank0_only": world > 1,
           "algorithmic_bytes_per_launch": int(alg), "algorithmic_gbs": round(alg / t / 1e9, 1), "pmc_record": key if rec else None,
           "pmc_file": ("profiles/" + f.name) if (rec and f is not None) else None}
    if rec:
        out["pmc_stale"] = rec.get("source_digest") != source_digest()
        out["pmc_kernel_ms"] = rec.get("kernel_ms")
    c = rec.get("pmc", {})
    traffic = rec.get("hbm_bytes_per_launch")
    insts = rec.get("SQ_INSTS_VALU") or c.get("SQ_INSTS_VALU")
    valu = insts / t / 1e9 if insts else None                      # G wave64 VALU instructions per second, whole chip
    fabric = traffic / t / 1e9 if traffic else None                # GB/s the L2 moved on its fabric side
    valu_frac = round(valu / VALU_PEAK_GINST, 4) if valu else None
    fabric_frac = round(fabric / HBM_PEAK_GBS, 4) if fabric else None
    mix = valu_mix_ceiling(rec, lds_resident) if rec else None
    out.update({"traffic": traffic, "valu_insts_per_launch": insts, "lane_util": rec.get("lane_util"), "lds_bank_conflict_frac": rec.get("lds_bank_conflict_frac"),
                "valu_frac": valu_frac, "l2_fabric_frac": fabric_frac})
    if mix:
        out["valu_mix_ceiling"] = mix                               # what this instruction mix can issue at most, as a fraction of the nominal peak
        if valu_frac:
            out["valu_frac_of_mix_ceiling"] = round(valu_frac / mix["frac_of_peak"], 4)
    # scenes read from global memory: the texture addresser (every wave-level load instruction costs it ~21-27 cycles whatever its width), the L1 <- L2 gather and how long
    # the waves sit in s_waitcnt -- from the gather passes of tools/profile_round.py (TA / TCP / TCC / SQ_WAIT counters, each pass its own run)
    ta_frac = gather_frac = None
    if c.get("TA_TA_BUSY_sum"):
        ta = c["TA_TA_BUSY_sum"] / t / 1e9                           # G TA-busy cycles per second, summed over the 256 addressers
        ta_frac = round(ta / TA_PEAK_GCYC, 4)
        out.update({"ta_busy_frac": ta_frac, "vmem_wave_instr_per_launch": c.get("TA_FLAT_READ_WAVEFRONTS_sum"),
                    "ta_cycles_per_load_instr": round(c["TA_TA_BUSY_sum"] / c["TA_FLAT_READ_WAVEFRONTS_sum"], 2) if c.get("TA_FLAT_READ_WAVEFRONTS_sum") else None})
    if c.get("TCP_TCC_READ_REQ_sum"):
        gbs = c["TCP_TCC_READ_REQ_sum"] * 128.0 / t / 1e9            # a request = one 128-byte line
        gather_frac = round(gbs / L2_GATHER_GBS, 4)
        out["l2_gather"] = {"achieved_gbs": round(gbs, 1), "roof_gbs": L2_GATHER_GBS, "frac": gather_frac, "requests_per_launch": c["TCP_TCC_READ_REQ_sum"],
                            "l1_hit": round(1.0 - c["TCP_TCC_READ_REQ_sum"] / c["TCP_TOTAL_CACHE_ACCESSES_sum"], 4) if c.get("TCP_TOTAL_CACHE_ACCESSES_sum") else None,
                            "l2_hit": round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4) if c.get("TCC_HIT_sum") else None}
    if c.get("SQ_WAIT_ANY") and c.get("SQ_WAVE_CYCLES"):
        out["wave_wait_frac"] = round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 4)       # share of the resident waves' cycles spent parked in s_waitcnt
    # the bound reported is the resource the kernel comes closest to saturating; VALU issue is compared through its mix ceiling when that is known
    cands = []
    if valu_frac is not None:
        cands.append(("valu", (valu_frac / mix["frac_of_peak"]) if mix else valu_frac, valu, VALU_PEAK_GINST, "G wave-instr/s", valu_frac))
    if ta_frac is not None:
        cands.append(("ta", ta_frac, c["TA_TA_BUSY_sum"] / t / 1e9, TA_PEAK_GCYC, "G addresser-cycles/s", ta_frac))
    if gather_frac is not None:
        cands.append(("l2_gather", gather_frac, c["TCP_TCC_READ_REQ_sum"] * 128.0 / t / 1e9, L2_GATHER_GBS, "GB/s", gather_frac))
    if fabric_frac is not None:
        cands.append(("l2_fabric", fabric_frac, fabric, HBM_PEAK_GBS, "GB/s", fabric_frac))
    if cands:
        b = max(cands, key=lambda x: x[1])
        out.update({"bound": b[0], "achieved": round(b[2], 1), "peak": b[3], "unit": b[4], "frac": round(b[5], 4)})
    else:
        out.update({"bound": "valu", "achieved": None, "peak": VALU_PEAK_GINST, "unit": "G wave-instr/s", "frac": None})
    if lds_resident:
        out["note"] = ("scene staged in LDS: the kernel is bound by VALU issue. achieved = SQ_INSTS_VALU (profiles/rNN_pmc.json) / kernel time measured in this run; peak = 1024 SIMD x 2.4 GHz / 2 cycles. "
                       "valu_mix_ceiling.frac_of_peak = what a stream with this kernel's dynamic class mix (SQ_INSTS_VALU_* counters) issues at most, measured (profiles/r04_measurements/valu_mix_rates.log); "
                       "lane_util says how many of the issued lanes did work; `traffic` is what the L2 moved on its fabric side")
    else:
        out["note"] = ("scene read from global memory (L2 / Infinity Cache resident). Three resources are close to their limits together and `bound` names the closest: VALU issue against the ceiling of "
                       "the kernel's instruction mix (valu_frac_of_mix_ceiling), the texture addresser (ta_busy_frac: it spends ~21-27 cycles per wave-level load instruction whatever the width, so "
                       "loads per ray is what counts, not bytes), and the L1 <- L2 line gather against the guide's 16.8-18.8 TB/s L2-hit figure (l2_gather, the upper one); wave_wait_frac = share of wave cycles in s_waitcnt")
    return out


# ------------------------------------------------------------------------------------------------------------------
# measurement
# ------------------------------------------------------------------------------------------------------------------

class Ctx:
    pass


def setup(args):
    import torch
    import torch.distributed as dist
    from terra_amd import runtime
    c = Ctx()
    c.world = int(os.environ.get("WORLD_SIZE", "1")); c.rank = int(os.environ.get("RANK", "0")); c.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # the multi-rank code path; TERRA_BENCH_DIST1=1 takes it with ONE rank too (shard, pack, RCCL gather, unpack): a self-test of
    # those calls on a 1-GPU box, never a benchmark
    c.dist_on = c.world > 1 or os.environ.get("TERRA_BENCH_DIST1") == "1"
    ngpu = torch.cuda.device_count()
    if ngpu < 1:
        raise SystemExit("bench.py needs an MI355X")
    dev_index = (c.local_rank % ngpu) if c.world > 1 else 0        # ranks > GPUs only happens in a gloo rehearsal
    torch.cuda.set_device(dev_index)
    if c.dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=c.rank, world_size=c.world, device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(args.dist_backend, rank=c.rank, world_size=c.world)
    c.dev = torch.device("cuda", dev_index)
    c.via_host = c.dist_on and args.dist_backend != "nccl"
    c.dist = dist; c.torch = torch
    c.lib = runtime.load()
    if c.lib.device_count() <= 0:
        raise SystemExit("bench.py needs an MI355X: " + runtime.last_error())
    runtime.check(c.lib.set_device(c.dev.index), "terra_amd_set_device")
    return c


def measure(c, d, tree, split, steps, warmup, check=False, prewarm_rect=None):
    """K timed steps of workload d; returns the measurement (rank 0) incl. per-launch work counters"""
    from terra_amd import runtime, scenes
    torch, dist, lib, dev = c.torch, c.dist, c.lib, c.dev
    world, rank = c.world, c.rank
    lib.clear_error()
    scene = scenes.build_scene(lib, d, tree_mode=TREE_MODES[tree], counters=False)      # the library's default: no work counters in the timed launches
    commit_ms = scenes.LAST_COMMIT_MS                                                    # terra_scene_commit alone: host tree build(s) + flattening + upload (reported separately)
    if runtime.last_error():
        raise SystemExit("scene commit failed: " + runtime.last_error())
    if getattr(c, "job_order", -1) >= 0:
        runtime.check(lib.set_job_order(scene, c.job_order), "terra_amd_set_job_order")
    ti = runtime.TraversalInfo(); runtime.check(lib.traversal_info(scene, C.byref(ti)))
    if split == 0:          # the library's automatic choice for THIS rank count's launches, as a number: every launch of the run (timed, counted, checked) then uses the same chunks
        split = lib.auto_sample_split(d.width, d.height, TILE, world, d.spp, int(bool(ti.lds_resident) and getattr(c, "job_order", -1) != 0))
        if split < 1:
            raise SystemExit("terra_amd_auto_sample_split: " + runtime.last_error())
    runtime.check(lib.set_sample_split(scene, split), "terra_amd_set_sample_split")
    info = runtime.SceneInfo(); runtime.check(lib.scene_info(scene, C.byref(info)))
    cam = scenes.camera_of(d)
    W, H = d.width, d.height
    cur = {"scene": scene, "fb": runtime.DeviceFramebuffer(W, H, device=dev)}      # what step() renders into (the check pass swaps both)
    main = torch.cuda.current_stream(dev)
    side = torch.cuda.Stream(dev) if c.dist_on else None
    n_packed = runtime.packed_floats_per_rank(W, H, TILE, world)
    packed = torch.zeros(n_packed, dtype=torch.float32, device=dev) if c.dist_on else None
    pool = [torch.zeros(n_packed, dtype=torch.float32, device="cpu" if c.via_host else dev) for _ in range(world)] if (c.dist_on and rank == 0) else []
    stage = torch.zeros(n_packed, dtype=torch.float32, device=dev) if (c.via_host and rank == 0) else None
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    PH = ("side_start", "packed", "gathered", "unpacked")
    pev = [{k: torch.cuda.Event(enable_timing=True) for k in PH} for _ in range(steps)] if c.dist_on else []      # phase marks on the side stream
    host_ph = [dict() for _ in range(steps)]                                                                      # ... and on the host clock (the gloo rehearsal's gather is host work)
    ev_packed = torch.cuda.Event()

    def fb_pack(r):          # on the side stream
        fb = cur["fb"]
        runtime.check(lib.pack_tiles(fb.pixels.data_ptr(), fb.results.data_ptr(), W, H, 0, 0, W, H, TILE, r, world, packed.data_ptr(), side.cuda_stream), "pack")
        ev_packed.record(side)
        return packed.cpu() if c.via_host else packed

    def fb_unpack(src, buf):
        fb = cur["fb"]
        if c.via_host:
            stage.copy_(buf); buf = stage
        runtime.check(lib.unpack_tiles(fb.pixels.data_ptr(), fb.results.data_ptr(), W, H, 0, 0, W, H, TILE, src, world, buf.data_ptr(), side.cuda_stream), "unpack")

    cursor = [0]
    def make_buffer(n):      # the gather's receive buffers, pre-allocated (one per rank)
        b = pool[cursor[0] % len(pool)]; cursor[0] += 1
        return b

    def step(i=None):
        fb, sc = cur["fb"], cur["scene"]
        if i is not None:
            ev[i][0].record(main)
        if not c.dist_on:
            runtime.check(lib.render_device(C.byref(cam), sc, fb.pixels.data_ptr(), fb.results.data_ptr(), W, H, 0, 0, W, H, None, main.cuda_stream), "render")
        else:
            runtime.check(lib.render_device_sharded(C.byref(cam), sc, fb.pixels.data_ptr(), fb.results.data_ptr(), W, H, 0, 0, W, H, TILE, rank, world, None, main.cuda_stream), "render")
        if i is not None:
            ev[i][1].record(main)
        if c.dist_on:
            side.wait_stream(main)                   # the pack reads what this render wrote
            with torch.cuda.stream(side):
                def mark(name):
                    if i is not None:
                        pev[i][name].record(side); host_ph[i][name] = time.perf_counter()
                mark("side_start")
                runtime.gather_frame(fb_pack, fb_unpack, W, H, TILE, rank, world, dist, make_buffer, mark=mark)
            main.wait_event(ev_packed)               # the next render may overwrite the rank's own tiles once they are packed; the gather and
                                                     # rank 0's unpack (other ranks' tiles) overlap with it

    def fence():
        if c.dist_on:
            dist.barrier()
        torch.cuda.synchronize(dev)

    if prewarm_rect:          # loads the kernel's code object with a small rectangle when a full warm-up step would take seconds
        x, y, w, h = prewarm_rect
        fb = cur["fb"]
        runtime.check(lib.render_device(C.byref(cam), scene, fb.pixels.data_ptr(), fb.results.data_ptr(), W, H, x, y, w, h, None, main.cuda_stream), "render")
        fence(); fb.clear()
    for _ in range(warmup):
        step()
    fence()
    runtime.check(lib.reset_stats(scene))
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    fence()
    elapsed = own_elapsed = time.perf_counter() - t0
    if c.dist_on:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    kernel_ms = sum(a.elapsed_time(b) for a, b in ev) / steps
    # the work counters (rays, nodes, tests, hits: what the roofline's algorithmic bytes are made of) come from ONE EXTRA, untimed launch with the counters
    # enabled (terra_amd_set_work_counters): they are instrumentation, off by default, and the timed launches above ran without them
    runtime.check(lib.set_work_counters(scene, 1), "terra_amd_set_work_counters")
    runtime.check(lib.reset_stats(scene))
    step(); fence()
    st = runtime.Stats(); runtime.check(lib.get_stats(scene, C.byref(st))); st = st.as_dict()
    runtime.check(lib.set_work_counters(scene, 0), "terra_amd_set_work_counters")
    launches = max(1, st["launches"])

    # per-rank phase breakdown (N > 1): where a step's time goes on every rank, so that a scaling shortfall can be attributed
    phases = None
    if c.dist_on:
        def avg(a, b, host=False):
            if host:
                return sum((h[b] - h[a]) * 1e3 for h in host_ph if a in h and b in h) / steps
            return sum(p[a].elapsed_time(p[b]) for p in pev) / steps
        mine = [own_elapsed / steps * 1e3, kernel_ms, avg("side_start", "packed"), avg("packed", "gathered"), avg("gathered", "unpacked"), avg("packed", "gathered", host=True)]
        vec = torch.tensor(mine, dtype=torch.float64, device=dev)
        allv = [torch.zeros_like(vec) for _ in range(world)]
        dist.all_gather(allv, vec)
        if rank == 0:
            rows = [[float(x) for x in v.tolist()] for v in allv]
            phases = {"unit": "ms per step, HIP events (render: main stream; pack / gather / unpack: second stream, overlapping the next step's render)",
                      "per_rank": [{"rank": r, "step_wall_ms": round(v[0], 3), "render_ms": round(v[1], 3), "pack_ms": round(v[2], 3), "gather_ms": round(v[3], 3),
                                    "unpack_ms": round(v[4], 3), "gather_host_ms": round(v[5], 3)} for r, v in enumerate(rows)],
                      "slowest_rank": max(range(world), key=lambda r: rows[r][0]),
                      "slowest_render_rank": max(range(world), key=lambda r: rows[r][1]),
                      "gather_bytes_to_rank0": int(n_packed * 4 * (world - 1))}

    ok = None
    if check:
        # one extra LOW-SPP step through the same shard -> pack -> gather -> unpack path into a fresh frame must equal an unsharded render bit for bit (rank 0)
        dd = scenes.SceneDesc(**{**d.__dict__, "spp": max(split, min(d.spp, 4 * split))})
        scene2 = scenes.build_scene(lib, dd, tree_mode=TREE_MODES[tree], counters=False)
        runtime.check(lib.set_sample_split(scene2, split), "terra_amd_set_sample_split")
        cur["scene"], cur["fb"] = scene2, runtime.DeviceFramebuffer(W, H, device=dev)
        fence(); step(); fence()
        if rank == 0:
            ref_fb = runtime.DeviceFramebuffer(W, H, device=dev)
            runtime.check(lib.render_device(C.byref(cam), scene2, ref_fb.pixels.data_ptr(), ref_fb.results.data_ptr(), W, H, 0, 0, W, H, None, main.cuda_stream), "render")
            torch.cuda.synchronize(dev)
            fb = cur["fb"]
            ok = bool(torch.equal(ref_fb.pixels.view(torch.int32), fb.pixels.view(torch.int32)) and torch.equal(ref_fb.results, fb.results))
            del ref_fb
        fence()
        lib.scene_destroy(scene2)
    ti2 = runtime.TraversalInfo(); runtime.check(lib.traversal_info(scene, C.byref(ti2)))      # after the launches: what the last call actually ran
    out = {"elapsed": elapsed, "kernel_ms": kernel_ms, "check": ok, "phases": phases, "commit_ms": commit_ms, "split": split,
           "per_launch": {k: v // launches for k, v in st.items() if k != "launches"},
           "traversal": {"fast_tree": bool(ti.fast_tree), "leaf_cull": bool(ti.leaf_cull), "note": ti.note.decode(), "last_call": getattr(ti2, "last_call", None)},
           "lds_resident": bool(ti.lds_resident),
           "triangles": info.triangles}
    lib.scene_destroy(scene)
    cur.clear()
    return out


def result_block(d, name, tree, split, steps, warmup, world, m, spp_override, integrator_name):
    frame_samples = d.width * d.height * d.spp
    st = m["per_launch"] or {}
    # the traversal the render calls actually ran (TerraAmdTraversalInfo::last_call: the commit-time decision can be overridden per call by the camera position)
    from terra_amd import runtime
    lc = m["traversal"].get("last_call")
    if lc in runtime.CALL_TRAVERSAL and lc:
        trav = runtime.CALL_TRAVERSAL[lc]
    else:
        trav = ("fast tree + reachability replay" if "reachability" in m["traversal"]["note"] else "fast tree") if m["traversal"]["fast_tree"] else ("reference tree + leaf-box cull" if m["traversal"]["leaf_cull"] else "reference tree, replica traversal")
    out = {"value": round(frame_samples * steps / m["elapsed"] / 1e6, 2), "unit": "Msamples/s", "steps": steps, "warmup": warmup,
           "ms_per_step": round(m["elapsed"] / steps * 1e3, 3), "commit_ms": round(m["commit_ms"], 2),
           "config": {"workload": name, "scene": d.name, "width": d.width, "height": d.height, "spp": d.spp, "bounces": d.bounces, "integrator": integrator_name,
                      "triangles": d.triangle_count, "tree": tree, "traversal": trav, "tile": TILE, "sample_split": split,
                      "parallelism": f"tiles%{world}" if world > 1 else "single"}}
    if st:
        out["mrays_per_s"] = round(st["rays"] * world / (m["kernel_ms"] * 1e-3) / 1e6, 1)
        out["roofline"] = roofline(pmc_key(name, tree, integrator_name, split, spp_override), st, m["kernel_ms"], m["lds_resident"], world)
        out["counters_per_launch"] = st
        out["counters_note"] = "from one extra untimed launch with terra_amd_set_work_counters(scene, 1); the timed launches run without work counters (the library's default)"
    return out


def host_api(c, d, tree, split, steps):
    """The same frame through the drop-in boundary itself: terra_render() (include/Terra.h:229) on a HOST framebuffer, so every call
    uploads the rectangle's running sums and brings pixels + sums back (44 B/pixel over PCIe) -- SURVEY.md 8(d)'s definition of the
    metric ("kernel + D2H of the tile included"). Two callers: one full-frame call per step, and the reference client's own pattern:
    128-pixel tiles (satellite/include/Config.hpp:25) dealt to 8 worker threads that call terra_render() concurrently
    (satellite/src/Renderer.cpp:70-98,316-350), with the sample split chosen per call (terra_amd_set_sample_split 0) so that a
    tile-sized call still fills the GPU. Reported beside `value`, never as it."""
    from terra_amd import api, runtime, scenes
    lib = c.lib
    lib.clear_error()
    scene = scenes.build_scene(lib, d, tree_mode=TREE_MODES[tree], counters=False); cam = scenes.camera_of(d)
    fb = api.Framebuffer(lib, d.width, d.height)              # terra_framebuffer_create: pinned host memory
    samples = d.width * d.height * d.spp
    out = {"pcie_bytes_per_step": d.width * d.height * 44, "steps": steps}

    def timed(fn):
        fn()                                                   # warm-up (stream + staging buffer of the calling threads)
        t = time.perf_counter()
        for _ in range(steps):
            fn()
        return (time.perf_counter() - t) / steps

    runtime.check(lib.set_sample_split(scene, split))
    dt = timed(lambda: lib.render(C.byref(cam), scene, C.byref(fb.fb), 0, 0, d.width, d.height))
    out["full_frame_call"] = {"ms_per_step": round(dt * 1e3, 3), "value": round(samples / dt / 1e6, 2), "unit": "Msamples/s", "sample_split": split}
    tiles = [(x, y, min(128, d.width - x), min(128, d.height - y)) for y in range(0, d.height, 128) for x in range(0, d.width, 128)]
    runtime.check(lib.set_sample_split(scene, 0))

    # 8 persistent workers, as the client's job system keeps them (satellite/src/Renderer.cpp:70-98): a worker's stream and staging buffers live as long as it does
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max_workers=8)

    def tile_loop():
        def worker(k):
            for t in tiles[k::8]:
                lib.render(C.byref(cam), scene, C.byref(fb.fb), *t)
        list(pool.map(worker, range(8)))
    dt = timed(tile_loop)
    pool.shutdown()
    out["tile_loop_128px_8_threads"] = {"ms_per_step": round(dt * 1e3, 3), "value": round(samples / dt / 1e6, 2), "unit": "Msamples/s", "tiles": len(tiles), "sample_split": "automatic per call"}
    err = runtime.last_error()
    buf = C.create_string_buffer(256)
    if lib.first_error(buf, 256) != 0:
        err = err or buf.value.decode()
    if err:
        out["error"] = err
    assert (fb.results["samples"] == d.spp * 2 * (steps + 1)).all(), "terra_render(): every pixel must have received every step's samples"
    fb.destroy(); lib.scene_destroy(scene)
    return out


def multi_device_block(workload_name, limit_s=300.0):
    """The in-process multi-GPU path of the C-ABI (terra_amd_set_devices + terra_amd_render_multi: tiles dealt to the devices from ONE process, one RCCL gather issued by
    the library) on 1, 2, 4, ... of the devices THIS process sees -- run as a child (tools/multi_device_bench.py) with a time limit, so that a failure or a hang of a device
    count that has never run before cannot take the bench line with it. On a one-GPU box this is the one-device self-test of that path."""
    cmd = [sys.executable, str(ROOT / "tools" / "multi_device_bench.py"), "--workload", workload_name, "--steps", "2"]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=limit_s)
    except subprocess.TimeoutExpired:
        return {"error": f"tools/multi_device_bench.py did not finish within {limit_s:.0f} s"}
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not line:
        return {"error": f"tools/multi_device_bench.py exited with code {r.returncode}: " + (r.stderr or "")[-400:]}
    blk = json.loads(line[-1])
    blk["note"] = ("terra_amd_render_multi() on a pinned host framebuffer, PCIe included, from one process; more than one device has only ever run where this line says so "
                   "(`runs` lists the device counts that ran here)")
    return blk


EXTRA_WORKLOADS = [
    # (workload, tree, integrator, split, steps, warmup, prewarm rectangle, CPU seconds)
    # (sample split: with the job queue a launch wants >= ~20 jobs per resident lane, or its last jobs ramp down alone: hall 256 spp split 1 / 4 / 8 -> 276.6 / 255.1 / 251.0 ms)
    ("hall_1080p_256spp", "auto", "simple", 32, 2, 1, None, 8.0),                       # configs[2] on the default (automatic) path: fast tree
    ("hall_1080p_256spp", "reference", "simple", 32, 1, 0, (896, 476, 128, 128), 0.0),  # ... and the reference's own tree, replica traversal (5 s per step)
    ("hall_x100_1080p_64spp", "auto", "simple", 4, 2, 1, None, 6.0),                   # the hall outside the coordinate range: fast tree + reachability replay
    ("spheres_1080p_1024spp", "auto", "simple", 32, 1, 1, None, 8.0),                   # configs[3]: PARITY UNPINNED presets
    ("cornell_1080p_512spp_direct", "auto", "direct", DEFAULT_SPLIT, 3, 1, None, 8.0),             # configs[1] with the reference client's default integrator
    ("hall_1080p_64spp_direct", "auto", "direct", 8, 2, 1, None, 6.0),                             # configs[2]'s scene with that integrator (64 spp)
]


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args, argv))
    import torch  # noqa: F401  before the library: it must bind to the HIP runtime torch loads
    c = setup(args)
    c.job_order = args.job_order
    world, rank = c.world, c.rank
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE is {world}")

    d = workload(args.workload, args.spp)
    if args.integrator:
        d.integrator = INTEGRATORS[args.integrator]
    if args.bounces >= 0:
        d.bounces = args.bounces
    integ_name = {0: "simple", 1: "direct", 2: "mis"}.get(d.integrator, str(d.integrator))
    m = measure(c, d, args.tree, args.sample_split, args.steps, args.warmup, check=args.check or (world > 1 and not args.no_check))

    if rank == 0:
        blk = result_block(d, args.workload, args.tree, m["split"], args.steps, args.warmup, world, m, args.spp, integ_name)
        out = {"metric": "Msamples/s", "value": blk["value"], "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": blk["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": blk["config"]}
        for k in ("mrays_per_s", "roofline", "counters_per_launch"):
            if k in blk:
                out[k] = blk[k]
        out["commit_ms"] = blk["commit_ms"]          # terra_scene_commit (host tree build + upload) of this workload, not part of `value`
        out["traversal_note"] = m["traversal"]["note"]
        if m["check"] is not None:
            out["sharded_equals_unsharded"] = m["check"]
        if c.dist_on:
            out["dist_backend"] = args.dist_backend
        if m.get("phases"):
            out["phases"] = m["phases"]
        if world == 1 and not c.dist_on and not args.no_host_api:
            out["host_api"] = host_api(c, d, args.tree, m["split"], max(1, min(args.steps, 3)))
            # SURVEY.md 8(d)'s own definition of the metric -- wall time of terra_render(), kernel + the tile's D2H included -- as a top-level key beside `value`
            # (which is the device-resident rate the bench contract asks for)
            out["value_terra_render"] = out["host_api"]["full_frame_call"]["value"]
            out["host_api"]["multi_device"] = multi_device_block(args.workload)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(d)
        headline = args.workload == "cornell_1080p_512spp" and not (args.spp or args.integrator or args.bounces >= 0) and args.tree == "auto" and args.sample_split == DEFAULT_SPLIT
        if world == 1 and not c.dist_on and headline and not args.no_workloads:
            extra = []; cpu_cache = {}
            for name, tree, integ, split, steps, warmup, prewarm, cpu_s in EXTRA_WORKLOADS:
                dw = workload(name)
                mw = measure(c, dw, tree, split, steps, warmup, prewarm_rect=prewarm)
                b = result_block(dw, name, tree, mw["split"], steps, warmup, 1, mw, 0, integ)
                if name.startswith("spheres"):
                    b["parity"] = "unpinned: the GGX and glass presets are this repo's definitions (the reference's are dead code, src/TerraPresets.c:298-465)"
                if not args.no_cpu_baseline:          # the CPU renders the scene, not the tree mode: one measurement per workload
                    if name not in cpu_cache and cpu_s:
                        cpu_cache[name] = cpu_baseline(dw, cpu_s)
                    if name in cpu_cache:
                        b["cpu_baseline"] = cpu_cache[name]
                extra.append(b)
            out["workloads"] = extra
        print(json.dumps(out), flush=True)
    if c.dist_on:
        c.dist.barrier()
        c.dist.destroy_process_group()


if __name__ == "__main__":
    main()
