"""The N>1 path on CPU: two processes (gloo), each renders its tile shard with the
ORACLE (no GPU here), packs its tiles in the product's packed layout, one gather to
rank 0, unpack; rank 0's frame must equal a single-process full-frame render.
Exercises terra_amd.runtime's shard/gather logic that bench.py drives over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def pack_np(pixels, results, tile, rank, world):
    """numpy statement of terra_amd_pack_tiles' layout: per tile, tile^2 pixels (3 floats) then tile^2 results (4 words)"""
    H, W, _ = pixels.shape
    tx, ty = -(-W // tile), -(-H // tile)
    most = -(-(tx * ty) // world)
    out = np.zeros((most, tile * tile * 7), np.float32)
    res_words = results.view(np.float32).reshape(H, W, 4)
    k = 0
    for t in range(tx * ty):
        if t % world != rank:
            continue
        x0, y0 = (t % tx) * tile, (t // tx) * tile
        p = np.zeros((tile, tile, 3), np.float32); r = np.zeros((tile, tile, 4), np.float32)
        h, w = min(tile, H - y0), min(tile, W - x0)
        p[:h, :w] = pixels[y0:y0 + h, x0:x0 + w]; r[:h, :w] = res_words[y0:y0 + h, x0:x0 + w]
        out[k, : tile * tile * 3] = p.ravel(); out[k, tile * tile * 3:] = r.ravel()
        k += 1
    return out.ravel()


def unpack_np(pixels, results, tile, rank, world, packed):
    H, W, _ = pixels.shape
    tx, ty = -(-W // tile), -(-H // tile)
    res_words = results.view(np.float32).reshape(H, W, 4)
    packed = packed.reshape(-1, tile * tile * 7)
    k = 0
    for t in range(tx * ty):
        if t % world != rank:
            continue
        x0, y0 = (t % tx) * tile, (t // tx) * tile
        h, w = min(tile, H - y0), min(tile, W - x0)
        pixels[y0:y0 + h, x0:x0 + w] = packed[k, : tile * tile * 3].reshape(tile, tile, 3)[:h, :w]
        res_words[y0:y0 + h, x0:x0 + w] = packed[k, tile * tile * 3:].reshape(tile, tile, 4)[:h, :w]
        k += 1


def _worker(rank, world, port, out_path):
    import ctypes as C
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import harness as H
    from terra_amd import api, runtime, scenes
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tile = 32
    d = scenes.cornell_box(100, 70, 2, integrator=1)
    orc = H.lib("orc")
    f = orc.fn("orc_render_pixels", None, H.RENDER_PIXELS_SIG)
    scene = scenes.build_scene(orc, d); cam = scenes.camera_of(d)
    fb = api.Framebuffer(orc, d.width, d.height)
    tx = -(-d.width // tile)
    for t in runtime.shard_tiles(d.width, d.height, tile, rank, world):
        x0, y0 = (t % tx) * tile, (t // tx) * tile
        f(C.byref(cam), scene, C.byref(fb.fb), x0, y0, min(tile, d.width - x0), min(tile, d.height - y0), scenes.FRAME_SEED, None)
    pixels, results = fb.pixels, fb.results
    runtime.gather_frame(
        fb_pack=lambda r: torch.from_numpy(pack_np(pixels, results, tile, r, world)),
        fb_unpack=lambda src, buf: unpack_np(pixels, results, tile, src, world, buf.numpy()),
        width=d.width, height=d.height, tile=tile, rank=rank, world=world, dist=dist,
        make_buffer=lambda n: torch.zeros(n, dtype=torch.float32))
    if rank == 0:
        np.savez(out_path, pixels=pixels, samples=results["samples"])
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_gather_equals_single_process(H, orc_lib, tmp_path):
    import torch.multiprocessing as mp
    from terra_amd import scenes
    out = tmp_path / "rank0.npz"
    mp.spawn(_worker, args=(2, _free_port(), str(out)), nprocs=2, join=True)
    got = np.load(out)
    want = H.Unit("orc").render_pixels(scenes.cornell_box(100, 70, 2, integrator=1), want_calls=False)
    assert H.same_bits(got["pixels"], want["pixels"]) and np.array_equal(got["samples"], want["samples"])


def test_pack_layout_roundtrip():
    r = np.random.RandomState(0)
    from terra_amd import api
    px = r.uniform(size=(70, 100, 3)).astype(np.float32)
    res = np.zeros((70, 100), api.RESULT_DTYPE); res["acc"] = r.uniform(size=(70, 100, 3)); res["samples"] = r.randint(0, 9, (70, 100))
    px2 = np.zeros_like(px); res2 = np.zeros_like(res)
    for rank in range(3):
        unpack_np(px2, res2, 32, rank, 3, pack_np(px, res, 32, rank, 3))
    assert np.array_equal(px, px2) and np.array_equal(res, res2)


def _worker_one(rank, world, port, out_path):
    """a process group of ONE (bench.py's TERRA_BENCH_DIST1 self-test): gather_frame must issue the collective, not return early"""
    import torch
    import torch.distributed as dist
    from terra_amd import runtime
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = []
    real = dist.gather

    class Counting:          # torch.distributed with gather counted
        def __getattr__(self, name):
            return getattr(dist, name)

        def gather(self, *a, **k):
            calls.append(1); return real(*a, **k)
    n = runtime.packed_floats_per_rank(100, 70, 32, 1)
    mine = torch.arange(n, dtype=torch.float32)
    bufs = []
    marks = []
    runtime.gather_frame(fb_pack=lambda r: mine, fb_unpack=lambda src, buf: marks.append(("unpack", src)), width=100, height=70, tile=32, rank=0, world=1, dist=Counting(),
                         make_buffer=lambda k: bufs.append(torch.zeros(k, dtype=torch.float32)) or bufs[-1], mark=marks.append)
    ok = len(calls) == 1 and len(bufs) == 1 and torch.equal(bufs[0], mine) and marks == ["packed", "gathered", "unpacked"]      # rank 0's own tiles are never unpacked
    # without a process group (dist = None): the early return, nothing gathered
    marks2 = []
    runtime.gather_frame(fb_pack=lambda r: mine, fb_unpack=None, width=100, height=70, tile=32, rank=0, world=1, dist=None, make_buffer=None, mark=marks2.append)
    ok = ok and marks2 == ["packed", "gathered", "unpacked"]
    open(out_path, "w").write("ok" if ok else f"calls={len(calls)} bufs={len(bufs)} marks={marks} marks2={marks2}")
    dist.destroy_process_group()


def test_group_of_one_still_issues_the_gather(tmp_path):
    import torch.multiprocessing as mp
    out = tmp_path / "one.txt"
    mp.spawn(_worker_one, args=(1, _free_port(), str(out)), nprocs=1, join=True)
    assert out.read_text() == "ok"
