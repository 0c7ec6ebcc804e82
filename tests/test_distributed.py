"""World-size-2 CPU rehearsal (gloo) of the multi-GPU path: band partition + gather on rank 0 + interleave.
The per-rank buffers are synthetic functions of the global pixel so the assembled image is checkable exactly."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
bands = importlib.import_module("metal-pathtracer-arm64_amd.bands")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _pattern(height, width):
    y, x = np.meshgrid(np.arange(height), np.arange(width), indexing="ij")
    return np.stack([y * 1000.0 + x, y - x * 0.5, (y * width + x) % 7], axis=-1).astype(np.float32)


def _local_buffer(height, width, rank, world):
    full = _pattern(height, width)
    rows = bands.max_band_count(height, world) * bands.BAND_ROWS
    buf = np.zeros((rows, width, 3), dtype=np.float32)
    for b in range(bands.band_count(height, rank, world)):
        g = rank + b * world
        R = bands.BAND_ROWS
        y0, y1 = g * R, min(g * R + R, height)
        buf[b * R:b * R + (y1 - y0)] = full[y0:y1]
    return torch.from_numpy(buf)


def _worker(rank, world, port, height, width, result_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        local = _local_buffer(height, width, rank, world)
        img = bands.gather_bands(local, height, rank, world)
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)      # the max-over-ranks timing reduction bench.py uses
        assert t.item() == float(world)
        if rank == 0:
            np.save(result_path, img.numpy())
        else:
            assert img is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("height,width", [(1080, 64), (40, 24)])
def test_two_rank_gather_reassembles_the_image(tmp_path, height, width):
    port = _free_port()
    out = str(tmp_path / "img.npy")
    mp.spawn(_worker, args=(2, port, height, width, out), nprocs=2, join=True)
    img = np.load(out)
    assert img.shape == (height, width, 3)
    assert np.array_equal(img, _pattern(height, width))


def test_assemble_matches_numpy_reference_for_many_partitions():
    pt = importlib.import_module("metal-pathtracer-arm64_amd")
    for height in (16, 100, 1080):
        for world in (1, 2, 3, 8):
            parts = [_local_buffer(height, 8, r, world) for r in range(world)]
            img = bands.assemble(parts, height).numpy()
            assert np.array_equal(img, _pattern(height, 8))
            trimmed = [p.numpy()[: bands.band_count(height, r, world) * bands.BAND_ROWS] for r, p in enumerate(parts)]
            assert np.array_equal(pt.assemble_bands(trimmed, 8, height), img)
            assert [bands.band_count(height, r, world) for r in range(world)] == [pt.band_count(height, r, world) for r in range(world)]


def _geometry_worker(rank, world, port, scene_path, scenes_dir, cache_path, result_path):
    """bench.py's setup with N > 1: rank 0 prepares the geometry once (host only) and the others find the finished file after the
    barrier - never half a file (it is written under a temporary name and renamed)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pt = importlib.import_module("metal-pathtracer-arm64_amd")
        host = pt.HostScene.load(scene_path, scenes_dir)
        if rank == 0:
            seconds = pt.prepare_geometry(host.desc, cache_path)
            assert seconds > 0.0
        dist.barrier()
        assert os.path.exists(cache_path) and not os.path.exists(cache_path + ".tmp")
        size = os.path.getsize(cache_path)
        sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(sizes, torch.tensor([size], dtype=torch.int64))
        assert all(int(s.item()) == size for s in sizes)
        if rank == 1:
            with open(result_path, "w") as f:
                f.write(str(size))
    finally:
        dist.destroy_process_group()


def test_two_ranks_share_one_prepared_geometry(tmp_path):
    scenes = os.path.join(ROOT, "scenes")
    cache = str(tmp_path / "geometry.bin")
    out = str(tmp_path / "size.txt")
    mp.spawn(_geometry_worker, args=(2, _free_port(), os.path.join(ROOT, "tests", "golden", "cornell_small_mesh.scene"), scenes, cache, out), nprocs=2, join=True)
    assert int(open(out).read()) > 10000
    # the file is a function of the scene alone: a second preparation writes the same bytes (the builder is deterministic)
    pt = importlib.import_module("metal-pathtracer-arm64_amd")
    host = pt.HostScene.load(os.path.join(ROOT, "tests", "golden", "cornell_small_mesh.scene"), scenes)
    again = str(tmp_path / "again.bin")
    pt.prepare_geometry(host.desc, again)
    a, b = open(cache, "rb").read(), open(again, "rb").read()
    # (the header carries the three phase timings of the preparation: bytes 0..7 magic, 8..15 fingerprint; compare those and the arrays)
    assert a[:16] == b[:16] and len(a) == len(b) and a[256:] == b[256:]
