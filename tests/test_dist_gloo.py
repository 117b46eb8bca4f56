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


def _frame_value(k, px, py):
    """What "frame k" holds at pixel (px, py): different for every frame, so a shard buffer read too early or too late shows."""
    return np.stack([px * 1.0 + 0.25 + 1000.0 * k, py * 2.0 + 0.5 - 7.0 * k, ((px * 31 + py * 17 + k) % 255) / 255.0], axis=1)


def _pipeline_worker(rank, world, port, hsize, vsize, n_frames, depth, n_buffers, out_path):
    """bench.py's N > 1 loop with shards in flight (rt_amd.shard_pipeline), rehearsed on the CPU: "rendering" a shard is
    filling its buffer — with garbage at the begin, with the frame's values at the end, which is when the real render is
    complete — the gather is gloo's, asynchronous like the one on bench.py's communication stream, and the root scatters
    and checks every frame as the gather behind it is waited for."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cd = rt.Camera(hsize, vsize, math.pi / 3).desc()
    counts = [int(rt.lib.rt_shard_pixel_count(C.byref(cd), r, world)) for r in range(world)]
    n_max = max(counts)
    px, py, valid = rt.shard_pixels(cd, rank, world)
    shards = [torch.zeros((n_max, 3), dtype=torch.float64) for _ in range(n_buffers)]
    landed = [torch.zeros((world, n_max, 3), dtype=torch.float64) if rank == 0 else None for _ in range(n_buffers)]
    pending = [None] * n_buffers      # per shard buffer: (frame, the gather that reads it)
    checked = []

    def retire(b):
        if pending[b] is None:
            return
        k, work = pending[b]
        work.wait()
        if rank == 0:
            frame = rt.assemble_host(cd, world, [landed[b][r].numpy()[: counts[r]] for r in range(world)])
            yy, xx = np.mgrid[0:vsize, 0:hsize]
            expect = _frame_value(k, xx.reshape(-1), yy.reshape(-1)).reshape(vsize, hsize, 3)
            expect[vsize - 1, :, :] = 0.0
            expect[:, hsize - 1, :] = 0.0
            assert np.array_equal(frame, expect), (k, b)
            checked.append(k)
        pending[b] = None

    in_flight = {}
    for op, k, b in rt.shard_pipeline(n_frames, depth, n_buffers):
        if op == "begin":
            retire(b)                              # the gather that last read this buffer
            assert b not in in_flight.values() and len(in_flight) < depth + 1
            shards[b].fill_(float("nan"))          # being rendered: nothing may read it now
            in_flight[k] = b
        elif op == "end":
            assert in_flight.pop(k) == b
            shards[b].zero_()
            shards[b][: len(px)][valid] = torch.from_numpy(_frame_value(k, px[valid], py[valid]))
        else:
            assert k not in in_flight and pending[b] is None
            work = dist.gather(shards[b], list(landed[b].unbind(0)) if rank == 0 else None, dst=0, async_op=True)
            pending[b] = (k, work)
    for b in range(n_buffers):
        retire(b)
    if rank == 0:
        np.save(out_path, np.array(sorted(checked)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,depth,n_buffers,n_frames", [(2, 3, 6, 17), (3, 2, 4, 9), (2, 1, 2, 5), (2, 3, 4, 2)])
def test_shards_in_flight_schedule_gathers_every_frame_intact(tmp_path, world, depth, n_buffers, n_frames):
    out = str(tmp_path / "checked.npy")
    port = 29700 + (os.getpid() % 2000) + world * 7 + depth
    mp.spawn(_pipeline_worker, args=(world, port, 67, 35, n_frames, depth, n_buffers, out), nprocs=world, join=True)
    assert np.array_equal(np.load(out), np.arange(n_frames))


def test_shard_pipeline_needs_a_free_buffer():
    with pytest.raises(ValueError):
        list(rt.shard_pipeline(4, 3, 3))
    steps = list(rt.shard_pipeline(5, 2, 3))
    assert [s for s in steps if s[0] == "begin"] == [("begin", k, k % 3) for k in range(5)]
    assert [s[1] for s in steps if s[0] == "post"] == list(range(5)) and [s[1] for s in steps if s[0] == "end"] == list(range(5))
