"""world_size-2 (and 3) gloo rehearsal of the multi-GPU plumbing bench.py uses: tile-shard the frame, gather the
shards to rank 0 (one padded buffer, strided), assemble.  Runs on CPU: the shards are filled with a function of the pixel instead of being
rendered, so what is tested is the partition (rt_shard_pixel_count / slot->pixel map) and the collective."""
import ctypes as C
import math
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import rt_amd as rt


def _value(px, py):
    return np.stack([px * 1.0 + 0.25, py * 2.0 + 0.5, (px * 31 + py * 17) % 255 / 255.0], axis=1)


def _worker(rank, world, port, hsize, vsize, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cd = rt.Camera(hsize, vsize, math.pi / 3).desc()
    counts = [int(rt.lib.rt_shard_pixel_count(C.byref(cd), r, world)) for r in range(world)]
    n_max = max(counts)
    px, py, valid = rt.shard_pixels(cd, rank, world)
    shard = np.zeros((n_max, 3))
    shard[: len(px)][valid] = _value(px[valid], py[valid])
    t = torch.from_numpy(shard)
    # as bench.py: the root gathers into views of ONE [world, n_max, 3] buffer (what rt_assemble_strided reads in place)
    gathered_all = torch.zeros((world, n_max, 3), dtype=torch.float64) if rank == 0 else None
    gathered = list(gathered_all.unbind(0)) if rank == 0 else None
    dist.gather(t, gathered, dst=0)
    if rank == 0:
        frame = rt.assemble_host(cd, world, [gathered_all[r].numpy()[: counts[r]] for r in range(world)])
        np.save(out_path, frame)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,hsize,vsize", [(2, 67, 35), (3, 130, 50), (2, 9, 9)])
def test_sharded_gather_reassembles_the_frame(tmp_path, world, hsize, vsize):
    out = str(tmp_path / "frame.npy")
    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, hsize, vsize, out), nprocs=world, join=True)
    frame = np.load(out)
    yy, xx = np.mgrid[0:vsize, 0:hsize]
    expect = _value(xx.reshape(-1), yy.reshape(-1)).reshape(vsize, hsize, 3)
    expect[vsize - 1, :, :] = 0.0   # F1: last row / column are not part of the domain
    expect[:, hsize - 1, :] = 0.0
    assert np.array_equal(frame, expect)


def test_every_pixel_belongs_to_exactly_one_shard():
    cd = rt.Camera(203, 101, 1.0).desc()
    for world in (1, 2, 4, 8):
        seen = np.zeros((101, 203), dtype=np.int32)
        for r in range(world):
            px, py, valid = rt.shard_pixels(cd, r, world)
            np.add.at(seen, (py[valid], px[valid]), 1)
        assert np.all(seen[:100, :202] == 1) and not seen[100].any() and not seen[:, 202].any()
