"""The one entry point that allocates nothing (rt_canvas_to_ppm_noalloc, include/rt_abi.h): same bytes as
rt_canvas_to_ppm on two calls in a row, the second with a smaller canvas (the static scratch keeps the first call's
row offsets), plus the size limit and the too-small-buffer case.  Collected last on purpose (test_zz_)."""
import ctypes as C

import numpy as np
import pytest

import rt_amd as rt
from rt_amd import World

pytestmark = pytest.mark.gpu


def _both(world, rgb, use_len_out):
    import torch
    v, h = rgb.shape[:2]
    scene = world.scene()
    t = torch.from_numpy(np.ascontiguousarray(rgb)).to("cuda:0")
    q_ref = torch.zeros(rgb.shape, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    ref = rt.canvas_to_ppm_device(scene, t.data_ptr(), h, v, q_ref.data_ptr())       # the existing entry point
    cap = len(ref) + 64
    q = torch.full(rgb.shape, 7, dtype=torch.uint8, device="cuda:0")
    text = torch.full((cap,), 0x5a, dtype=torch.uint8, device="cuda:0")
    length = torch.zeros(1, dtype=torch.int64, device="cuda:0")
    torch.cuda.synchronize()
    if use_len_out:
        rc = rt.lib.rt_canvas_to_ppm_noalloc(scene, t.data_ptr(), h, v, q.data_ptr(), text.data_ptr(), cap, length.data_ptr(), None)
        assert rc == 0, rt.last_error()
        # stream order: the scene's stream is not torch's, so wait for the device before reading
        torch.cuda.synchronize()
        n = int(length.item())
    else:
        n = rt.lib.rt_canvas_to_ppm_noalloc(scene, t.data_ptr(), h, v, q.data_ptr(), text.data_ptr(), cap, None, None)
        assert n >= 0, rt.last_error()
    got = bytes(text.cpu().numpy()[:n].tobytes())
    assert n == len(ref)
    assert got == ref
    assert bytes(text.cpu().numpy()[n:].tobytes()) == b"\x5a" * (cap - n)      # nothing written past the text
    assert np.array_equal(q.cpu().numpy(), q_ref.cpu().numpy())
    return ref


def test_noalloc_ppm_matches_rt_canvas_to_ppm_twice_second_call_smaller():
    w = World.default()
    rng = np.random.default_rng(11)
    big = rng.uniform(-0.2, 1.2, size=(61, 97, 3))
    big[rng.uniform(size=big.shape[:2]) < 0.03] = np.nan
    small = rng.uniform(-0.2, 1.2, size=(9, 31, 3))
    a = _both(w, big, use_len_out=False)
    b = _both(w, small, use_len_out=False)       # smaller input straight after: stale offsets past row 9 must not matter
    assert a != b
    assert b == rt.Image(small, None, None, None).to_ppm().encode()        # and both equal the host mirror (canvas.rs:33-98)
    assert a == rt.Image(big, None, None, None).to_ppm().encode()
    _both(w, big, use_len_out=True)              # length delivered in stream order, no host wait inside the call
    _both(w, small, use_len_out=True)


def test_noalloc_ppm_leaves_device_memory_where_it_was():
    """hipMemGetInfo before and after a run of calls (both ways of receiving the length): not a byte of device memory
    is taken.  The length read-back of the d_len_out == NULL form goes into a page-locked word the scene has held since
    its creation, not into pageable memory the runtime would stage."""
    import torch
    w = World.default()
    scene = w.scene()
    rgb = torch.rand((40, 50, 3), dtype=torch.float64, device="cuda:0")
    q = torch.zeros((40, 50, 3), dtype=torch.uint8, device="cuda:0")
    text = torch.zeros((40 * 50 * 12 + 64,), dtype=torch.uint8, device="cuda:0")
    length = torch.zeros(1, dtype=torch.int64, device="cuda:0")
    args = (scene, rgb.data_ptr(), 50, 40, q.data_ptr(), text.data_ptr(), text.numel())
    n0 = rt.lib.rt_canvas_to_ppm_noalloc(*args, None, None)     # first call: whatever is lazily set up happens here
    assert n0 > 0, rt.last_error()
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info(0)
    for _ in range(20):
        assert rt.lib.rt_canvas_to_ppm_noalloc(*args, None, None) == n0
        assert rt.lib.rt_canvas_to_ppm_noalloc(*args, length.data_ptr(), None) == 0
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info(0)
    assert free1 == free0, (free0, free1)
    assert int(length.item()) == n0


def test_noalloc_ppm_limits():
    import torch
    w = World.default()
    scene = w.scene()
    rgb = torch.zeros((4, 5, 3), dtype=torch.float64, device="cuda:0")
    q = torch.zeros((4, 5, 3), dtype=torch.uint8, device="cuda:0")
    text = torch.full((16,), 0x5a, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    # more rows than the static scratch holds: a clear error, nothing launched
    rc = rt.lib.rt_canvas_to_ppm_noalloc(scene, rgb.data_ptr(), 1, 16385, q.data_ptr(), text.data_ptr(), 16, None, None)
    assert rc == rt.RT_ERR_UNSUPPORTED and "16384" in rt.last_error()
    # the quantised frame is required
    rc = rt.lib.rt_canvas_to_ppm_noalloc(scene, rgb.data_ptr(), 5, 4, None, text.data_ptr(), 16, None, None)
    assert rc == rt.RT_ERR_BAD_ARG
    # a buffer that cannot hold the text: the length comes back, not one byte is written
    n = rt.lib.rt_canvas_to_ppm_noalloc(scene, rgb.data_ptr(), 5, 4, q.data_ptr(), text.data_ptr(), 16, None, None)
    assert n == len(rt.Image(np.zeros((4, 5, 3)), None, None, None).to_ppm().encode()) and n > 16
    assert bytes(text.cpu().numpy().tobytes()) == b"\x5a" * 16
