"""Host-side runtime helpers around libterra_amd.so.

PyTorch is used only as plumbing: device memory (framebuffers are torch tensors
whose data_ptr() is handed to the C-ABI), streams and torch.distributed (RCCL).
Import order matters on this image: torch ships its own libamdhip64.so.7, so
torch is imported BEFORE the library is dlopen'ed and the library binds to the
runtime torch already loaded (two HIP runtimes in one process do not see each
other's devices).
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path
from typing import Optional, Tuple

import numpy as np

from . import api, scenes

HERE = Path(__file__).resolve().parent
import os as _os
# 8 hardware queues instead of ROCm's default 4: terra_render() callers on 8 threads (the reference client's pattern) each own a stream, and streams that share a
# hardware queue run their kernels one after the other. Read by the HIP runtime at its initialisation, so it is set here, before anything touches the GPU; the
# library asks for the same when it is loaded on its own (scene_host.cpp). A value the user has set wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
LIB_PATH = Path(_os.environ.get("TERRA_AMD_LIB", HERE / "libterra_amd.so"))     # TERRA_AMD_LIB: experiment builds (terra_amd/build.py --variant)

_lib: Optional[api.TerraLib] = None


class TerraAmdError(RuntimeError):
    pass


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("rays", "nodes", "box_tests", "tri_tests", "hits", "samples", "rand_calls", "attr_fetches", "pixels", "launches", "tri_culled")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class TraversalInfo(C.Structure):
    _fields_ = [("tree_mode", C.c_int), ("fast_tree", C.c_int), ("fast_tree_built_on_device", C.c_int), ("leaf_cull", C.c_int), ("lds_resident", C.c_int), ("max_coordinate", C.c_float), ("max_coordinate_allowed", C.c_float), ("note", C.c_char * 256),
                ("last_call", C.c_int), ("camera_limit", C.c_float)]


CALL_TRAVERSAL = {0: "none", 1: "reference tree, replica traversal", 2: "reference tree + leaf-box cull", 3: "fast tree", 4: "fast tree + reachability replay"}


class SceneInfo(C.Structure):
    _fields_ = [("triangles", C.c_uint32), ("nodes", C.c_uint32), ("objects", C.c_uint32), ("lights", C.c_uint32),
                ("lights_triangles_count", C.c_uint32), ("max_stack", C.c_int32), ("device_bytes", C.c_uint64)]


_CAM = C.POINTER(api.TerraCamera)
_SZ = C.c_size_t
_EXTRA = {
    "terra_amd_last_error": (C.c_char_p, []),
    "terra_amd_clear_error": (None, []),
    "terra_amd_first_error": (C.c_int, [C.c_char_p, C.c_size_t]),
    "terra_amd_clear_first_error": (None, []),
    "terra_amd_thread_staging_bytes": (C.c_size_t, []),
    "terra_amd_device_count": (C.c_int, []),
    "terra_amd_set_device": (C.c_int, [C.c_int]),
    "terra_amd_get_device": (C.c_int, []),
    "terra_amd_set_frame_seed": (None, [C.c_void_p, C.c_uint64]),
    "terra_amd_debug_shrink_reference_boxes": (C.c_int, [C.c_void_p, C.c_float]),
    "terra_amd_debug_pad_stack": (C.c_int, [C.c_void_p, C.c_int]),
    "terra_amd_debug_fast_stack_lds": (C.c_int, [C.c_void_p, C.c_int]),
    "terra_amd_scene_supported": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t]),
    "terra_amd_init": (C.c_int, []),
    "terra_amd_set_commit_timing": (None, [C.c_int]),
    "terra_amd_set_build_threads": (C.c_int, [C.c_int]),
    "terra_amd_set_azimuth_table": (None, [C.c_int]),
    "terra_amd_set_devices": (C.c_int, [C.POINTER(C.c_int), C.c_int]),
    "terra_amd_get_devices": (C.c_int, [C.POINTER(C.c_int), C.c_int]),
    "terra_amd_shard_owner": (C.c_int, [C.c_size_t, C.c_int]),
    "terra_amd_render_multi": (C.c_int, [C.POINTER(api.TerraCamera), C.c_void_p, C.POINTER(api.TerraFramebuffer)] + [C.c_size_t] * 5),
    "terra_amd_multi_info": (C.c_int, [C.c_void_p, C.c_void_p]),
    "terra_amd_debug_replicas_share_device": (C.c_int, [C.c_int]),
    "terra_amd_set_work_counters": (C.c_int, [C.c_void_p, C.c_int]),
    "terra_amd_get_work_counters": (C.c_int, [C.c_void_p]),
    "terra_amd_set_sampler_integration": (C.c_int, [C.c_void_p, C.c_int]),
    "terra_amd_get_sampler_integration": (C.c_int, [C.c_void_p]),
    "terra_amd_set_environment_sampling": (C.c_int, [C.c_void_p, C.c_int]),
    "terra_amd_get_environment_sampling": (C.c_int, [C.c_void_p]),
    "terra_amd_get_frame_seed": (C.c_uint64, [C.c_void_p]),
    "terra_amd_set_tree_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "terra_amd_get_tree_mode": (C.c_int, [C.c_void_p]),
    "terra_amd_traversal_info": (C.c_int, [C.c_void_p, C.POINTER(TraversalInfo)]),
    "terra_amd_set_tree_builder": (C.c_int, [C.c_void_p, C.c_int]),
    "terra_amd_get_tree_builder": (C.c_int, [C.c_void_p]),
    "terra_amd_set_job_order": (C.c_int, [C.c_void_p, C.c_int]),
    "terra_amd_get_job_order": (C.c_int, [C.c_void_p]),
    "terra_amd_set_sample_split": (C.c_int, [C.c_void_p, C.c_int]),
    "terra_amd_get_sample_split": (C.c_int, [C.c_void_p]),
    "terra_amd_auto_sample_split": (C.c_int, [C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_size_t, C.c_int]),
    "terra_amd_set_environment_lighting": (C.c_int, [C.c_void_p, C.c_int]),
    "terra_amd_get_environment_lighting": (C.c_int, [C.c_void_p]),
    "terra_amd_get_stats": (C.c_int, [C.c_void_p, C.POINTER(Stats)]),
    "terra_amd_reset_stats": (C.c_int, [C.c_void_p]),
    "terra_amd_scene_info": (C.c_int, [C.c_void_p, C.POINTER(SceneInfo)]),
    "terra_amd_scene_bvh_nodes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "terra_amd_render_device": (C.c_int, [_CAM, C.c_void_p, C.c_void_p, C.c_void_p] + [_SZ] * 6 + [C.c_void_p, C.c_void_p]),
    "terra_amd_render_device_sharded": (C.c_int, [_CAM, C.c_void_p, C.c_void_p, C.c_void_p] + [_SZ] * 7 + [C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "terra_amd_shard_tile_count": (C.c_int, [_SZ, _SZ, _SZ, C.c_int, C.c_int]),
    "terra_amd_shard_packed_bytes": (_SZ, [_SZ, _SZ, _SZ, C.c_int]),
    "terra_amd_pack_tiles": (C.c_int, [C.c_void_p, C.c_void_p] + [_SZ] * 7 + [C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "terra_amd_unpack_tiles": (C.c_int, [C.c_void_p, C.c_void_p] + [_SZ] * 7 + [C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "terra_amd_synchronize": (C.c_int, [C.c_void_p]),
    "terra_amd_time_render_device": (C.c_int, [_CAM, C.c_void_p, C.c_void_p, C.c_void_p] + [_SZ] * 6 + [C.c_int, C.c_void_p, C.POINTER(C.c_float)]),
}


def load(need_torch: bool = True) -> api.TerraLib:
    """dlopen libterra_amd.so (after torch, see module docstring). Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if need_torch:
        import torch  # noqa: F401  (loads the HIP runtime the library will bind to)
    if not LIB_PATH.exists():
        raise TerraAmdError(f"{LIB_PATH} is missing: run `python -m terra_amd.build` (there is no CPU fallback)")
    lib = api.TerraLib(LIB_PATH, "terra_")
    if lib.missing:
        raise TerraAmdError(f"{LIB_PATH} lacks Terra.h entry points: {lib.missing}")
    for name, (res, args) in _EXTRA.items():
        if lib.has(name):          # (an older build of the library, loaded through TERRA_AMD_LIB for an A/B, lacks the newer entry points: calling one then fails by name)
            setattr(lib, name[len("terra_amd_"):] if name.startswith("terra_amd_") else name, lib.fn(name, res, args))
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> int:
    if rc < 0:
        raise TerraAmdError(f"{what}: {load().last_error().decode()} (status {rc})")
    return rc


def last_error() -> str:
    return load().last_error().decode()


def first_error():
    """(status, message) of the first error any thread recorded since the last clear_first_error(); (0, "") if none"""
    buf = C.create_string_buffer(512)
    st = load().first_error(buf, 512)
    return st, buf.value.decode()


class MultiInfo(C.Structure):
    """TerraAmdMultiInfo (include/terra_amd.h)"""
    _fields_ = [("devices", C.c_int), ("device", C.c_int * 16), ("replicas", C.c_int), ("gathers", C.c_uint64), ("last_gather_bytes", C.c_uint64),
                ("process_collectives", C.c_uint64), ("rccl_version", C.c_int), ("communicator_ranks", C.c_int), ("rccl_library", C.c_char * 64), ("rehearsed_gathers", C.c_uint64)]


class DeviceFramebuffer:
    """A TerraFramebuffer that lives in HBM: torch tensors, handed to the C-ABI as raw pointers."""

    def __init__(self, width: int, height: int, device: str = "cuda"):
        import torch
        self.width, self.height = width, height
        self.pixels = torch.zeros(height * width * 3, dtype=torch.float32, device=device)
        self.results = torch.zeros(height * width * 4, dtype=torch.int32, device=device)   # {f32 acc[3]; i32 samples}

    def clear(self):
        self.pixels.zero_(); self.results.zero_()

    def pixels_host(self) -> np.ndarray:
        return self.pixels.cpu().numpy().reshape(self.height, self.width, 3)

    def results_host(self) -> np.ndarray:
        return self.results.cpu().numpy().view(api.RESULT_DTYPE).reshape(self.height, self.width)


def render_device(lib, cam, scene, fb: DeviceFramebuffer, rect: Optional[Tuple[int, int, int, int]] = None, rand_calls=None, stream=None):
    x, y, w, h = rect if rect else (0, 0, fb.width, fb.height)
    check(lib.render_device(C.byref(cam), scene, fb.pixels.data_ptr(), fb.results.data_ptr(), fb.width, fb.height, x, y, w, h,
                            rand_calls.data_ptr() if rand_calls is not None else None, stream), "terra_amd_render_device")


def render_device_sharded(lib, cam, scene, fb: DeviceFramebuffer, tile: int, rank: int, world: int, stream=None):
    check(lib.render_device_sharded(C.byref(cam), scene, fb.pixels.data_ptr(), fb.results.data_ptr(), fb.width, fb.height,
                                    0, 0, fb.width, fb.height, tile, rank, world, None, stream), "terra_amd_render_device_sharded")


# ---------------------------------------------------------------------------
# tile sharding across ranks (one process per GPU) and the single gather
# ---------------------------------------------------------------------------

def shard_tiles(width: int, height: int, tile: int, rank: int, world: int):
    """tile ids (row-major in the frame) owned by `rank`: t % world == rank (same rule as the kernel)"""
    tx, ty = -(-width // tile), -(-height // tile)
    return [t for t in range(tx * ty) if t % world == rank]


def packed_floats_per_rank(width: int, height: int, tile: int, world: int) -> int:
    """floats in one rank's packed gather buffer (padded to rank 0's tile count): 7 floats = 28 B per pixel"""
    tx, ty = -(-width // tile), -(-height // tile)
    most = -(-(tx * ty) // world)
    return most * tile * tile * 7


def gather_frame(fb_pack, fb_unpack, width, height, tile, rank, world, dist, make_buffer, dst=0, mark=None):
    """One gather of every rank's packed tiles to `dst`, then unpack there.

    fb_pack(rank) -> 1-D float32 tensor holding this rank's tiles in the packed layout
    fb_unpack(src_rank, packed) writes rank src_rank's tiles into the destination frame (called on dst only)
    dist: torch.distributed (backend nccl == RCCL on the GPU box, gloo in CPU tests)
    mark(name): optional, called after each phase is QUEUED ("packed", "gathered", "unpacked") -- bench.py records stream events there
    """
    mark = mark or (lambda name: None)
    mine = fb_pack(rank)
    mark("packed")
    if world == 1 and not (dist is not None and dist.is_available() and dist.is_initialized()):
        mark("gathered"); mark("unpacked")
        return
    # (one rank WITH a process group -- bench.py's TERRA_BENCH_DIST1 self-test -- still issues the collective: the gather of a group of one)
    n = packed_floats_per_rank(width, height, tile, world)
    assert mine.numel() == n
    if rank == dst:
        bufs = [make_buffer(n) for _ in range(world)]
        dist.gather(mine, gather_list=bufs, dst=dst)
        mark("gathered")
        for src in range(world):
            if src != dst:
                fb_unpack(src, bufs[src])
    else:
        dist.gather(mine, gather_list=None, dst=dst)
        mark("gathered")
    mark("unpacked")
