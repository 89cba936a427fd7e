"""Tile partition + gather to rank 0 (the only collective on the path). The
N>1 path is exercised with world_size-2 gloo processes on CPU."""
import os
import sys

import numpy as np
import pytest

from raytracer_amd import dist as rdist


def test_partition_covers_every_pixel_once():
    for (w, h, ws, tw, th) in [(3840, 2176, 8, 64, 32), (992, 992, 3, 64, 32), (200, 200, 2, 64, 32), (96, 40, 5, 16, 8)]:
        cover = np.zeros((h, w), np.int32)
        for r in range(ws):
            for (x0, y0) in rdist.tiles_of_rank(w, h, r, ws, tw, th):
                cover[y0:y0 + th, x0:x0 + tw] += 1
        assert (cover == 1).all()
        sizes = [len(rdist.tiles_of_rank(w, h, r, ws, tw, th)) for r in range(ws)]
        assert max(sizes) - min(sizes) <= 1


def test_pack_unpack_round_trip():
    rng = np.random.default_rng(0)
    f = rng.random((200, 200, 4), dtype=np.float32)
    out = np.zeros_like(f)
    for r in range(3):
        tiles = rdist.tiles_of_rank(200, 200, r, 3)
        rdist.unpack_tiles_host(out, rdist.pack_tiles_host(f, tiles), tiles)
    assert (out == f).all()


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    h, w = 104, 200
    yy, xx = np.mgrid[0:h, 0:w]
    truth = np.stack([xx, yy, xx * yy, np.ones_like(xx)], axis=-1).astype(np.float32)
    frame = np.full((h, w, 4), -1.0, np.float32)
    for (x0, y0) in rdist.tiles_of_rank(w, h, rank, world):
        frame[y0:y0 + 32, x0:x0 + 64] = truth[y0:y0 + 32, x0:x0 + 64]
    out = rdist.gather_frame_host(frame, rank, world)
    if rank == 0:
        q.put(bool((out == truth).all()))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_world_size_2_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok
