"""Builds terra_amd/libterra_amd.so (HIP kernels + C-ABI host side) for gfx950 with hipcc.

In-tree build: the .so lands next to this file so it travels with the repo
snapshot to the GPU box. Flags that matter for parity (DESIGN.md "Bit-faithful
arithmetic"): no FMA contraction, IEEE f32 division/sqrt, no fast-math, f32
denormals kept (hipcc default).
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
OUT = HERE / "libterra_amd.so"
# (source, object name, extra flags): render_kernels.hip is compiled once per TERRA_TU value -- its kernel instances per template MODE -- so that the units build in parallel
SOURCES = [("scene_host.cpp", "scene_host.cpp", []), ("tree_build.cpp", "tree_build.cpp", []), ("multi_gpu.cpp", "multi_gpu.cpp", []),
           ("render_kernels.hip", "render_kernels.tu0.hip", ["-DTERRA_TU=0"]), ("render_kernels.hip", "render_kernels.tu1.hip", ["-DTERRA_TU=1"]),
           ("render_kernels.hip", "render_kernels.tu2.hip", ["-DTERRA_TU=2"]), ("render_kernels.hip", "render_kernels.tu3.hip", ["-DTERRA_TU=3"]),
           ("unit_kernels.hip", "unit_kernels.hip", []), ("tree_build_device.hip", "tree_build_device.hip", [])]
HEADERS = ["dev_types.h", "dev_math.h", "rng.h", "trace_device.h", "sampling_device.h", "kernels.h", "tree_build.h", "multi_gpu.h"]
ARCH = os.environ.get("TERRA_AMD_ARCH", "gfx950")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

FLAGS = [
    "-x", "hip", f"--offload-arch={ARCH}", "-std=c++17", "-O3", "-fPIC",
    "-ffp-contract=off", "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt",
    # no SLP vectorisation: on gfx950 a v_pk_mul/add_f32 issues at 0.57 G/s per SIMD against 0.96 for v_mul/v_add_f32 (profiles/r02_measurements/valu_rates.log),
    # so a packed pair barely beats two scalar instructions and the v_mov shuffles that build the pairs are pure loss: headline 82.6 -> 75.7 ms
    "-fno-slp-vectorize",
    "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", "-Wno-unused-variable",
]


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = False, extra_flags=(), variant: str = "") -> Path:
    """variant != "": an experiment build (extra_flags, e.g. -DTERRA_LEAF_CAP=4) written to libterra_amd_<variant>.so;
    select it at run time with TERRA_AMD_LIB=<path> (terra_amd/runtime.py)."""
    objdir = HERE / ("build" + ("_" + variant if variant else ""))
    objdir.mkdir(exist_ok=True)
    OUT = HERE / (f"libterra_amd_{variant}.so" if variant else "libterra_amd.so")
    deps_common = [CSRC / h for h in HEADERS] + [HERE.parent / "include" / h for h in ("Terra.h", "TerraMath.h", "TerraPresets.h", "terra_amd.h")] + [Path(__file__)]
    # a change of flags (experiment builds are re-run with different -D values) invalidates the objects too
    stamp = objdir / "flags.txt"
    flags_now = " ".join([*FLAGS, *extra_flags])
    objdir.mkdir(parents=True, exist_ok=True)
    if not stamp.exists() or stamp.read_text() != flags_now:
        force = True
    jobs = []
    for src, name, tu_flags in SOURCES:
        obj = objdir / (name + ".o")
        if force or _stale(obj, [CSRC / src] + deps_common):
            jobs.append((src, obj, tu_flags))

    def compile_one(job):
        src, obj, tu_flags = job
        cmd = [HIPCC, *FLAGS, *tu_flags, *extra_flags, "-c", str(CSRC / src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, flush=True)
        return obj

    if jobs:
        with ThreadPoolExecutor(max_workers=len(jobs)) as ex:
            list(ex.map(compile_one, jobs))
    stamp.write_text(flags_now)
    objs = [objdir / (name + ".o") for _, name, _ in SOURCES]
    if force or jobs or _stale(OUT, objs):
        cmd = [HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-Wl,-Bsymbolic-functions", "-o", str(OUT), *map(str, objs)]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return OUT


if __name__ == "__main__":
    # python -m terra_amd.build [--force] [--variant NAME -DFOO=1 ...]
    args = [a for a in sys.argv[1:] if a != "--force"]
    variant = ""
    if "--variant" in args:
        i = args.index("--variant"); variant = args[i + 1]; del args[i:i + 2]
    p = build(force="--force" in sys.argv, verbose=True, extra_flags=tuple(args), variant=variant)
    print(p)
