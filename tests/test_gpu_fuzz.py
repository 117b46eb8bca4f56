"""Fuzz parity: seeded random worlds (tests/scenes.py: random_scene) rendered through the C ABI on the GPU and by the
oracle.  Same bars as tests/test_gpu_parity.py: primary hit ids and flags bit-exact, colours within 1e-9, ray counts
of every class equal.  Pixels whose ray tree ran into a condition the reference panics on (RT_FLAG_* set, equal on
both sides) are held to the same bar: the device and the oracle take the same documented way out.

RT_FUZZ_SEEDS=n widens the run to seeds 0..n-1 (the default keeps the suite short); RT_FUZZ_SIZE=WxH."""
import os

import numpy as np
import pytest

import oracle_bridge as ob
import rt_amd as rt
import scenes

pytestmark = pytest.mark.gpu
COLOR_TOL = 1e-9
N_SEEDS = int(os.environ.get("RT_FUZZ_SEEDS", "48"))
W, H = (int(x) for x in os.environ.get("RT_FUZZ_SIZE", "64x40").split("x"))
BATCH = 8


def compare(seed, depth, builder=None):
    world, cam = (builder or scenes.random_scene)(seed, W, H)
    g = cam.render(world, depth)
    o = ob.render(world, cam, depth)
    problems = []
    if not np.array_equal(g.hit_id, o.hit_id):
        problems.append(f"{int((g.hit_id != o.hit_id).sum())} primary hit ids differ")
    # a bounded device list that overflows is re-traced with its lists in the HBM arena: the flag is never seen
    if ((g.flags & rt.RT_FLAG_LIST_OVERFLOW) != 0).any():
        problems.append(f"{int(((g.flags & rt.RT_FLAG_LIST_OVERFLOW) != 0).sum())} pixels with RT_FLAG_LIST_OVERFLOW")
    if not np.array_equal(g.flags, o.flags):
        problems.append(f"{int((g.flags != o.flags).sum())} flag bytes differ")
    err = np.abs(g.rgb - o.rgb)
    err[np.isnan(g.rgb) & np.isnan(o.rgb)] = 0.0
    err[np.isinf(g.rgb) & (g.rgb == o.rgb)] = 0.0
    # relative above 1: materials outside [0, 1] amplify colours to 1e7
    err = err / np.maximum(1.0, np.nan_to_num(np.abs(o.rgb), nan=1.0, posinf=1.0))
    worst = float(np.nanmax(err)) if err.size else 0.0
    if not worst < COLOR_TOL or np.isnan(err).any():
        problems.append(f"colour error {worst:.3e} ({int((err >= COLOR_TOL).any(axis=-1).sum())} pixels, "
                        f"{int(np.isnan(err).any(axis=-1).sum())} one-sided NaN)")
    gs, os_ = g.stats, o.stats
    if (gs.rays_primary, gs.rays_reflect, gs.rays_refract, gs.rays_shadow, gs.shaded_nodes) != \
            (os_.rays_primary, os_.rays_reflect, os_.rays_refract, os_.rays_shadow, os_.shaded_nodes):
        problems.append("ray counts differ")
    if g.rgb[-1].any() or g.rgb[:, -1].any():
        problems.append("F1 border not black")
    return problems


@pytest.mark.parametrize("first", range(0, N_SEEDS, BATCH))
def test_random_worlds(first):
    failures = {}
    for seed in range(first, min(first + BATCH, N_SEEDS)):
        problems = compare(seed, depth=4 if seed % 3 else 6)
        if problems:
            failures[seed] = problems
    assert not failures, failures


@pytest.mark.parametrize("first", range(0, N_SEEDS, BATCH))
def test_random_wild_worlds(first):
    failures = {}
    for seed in range(first, min(first + BATCH, N_SEEDS)):
        problems = compare(seed, depth=4, builder=lambda s, w, h: scenes.random_scene(s, w, h, wild=True))
        if problems:
            failures[seed] = problems
    assert not failures, failures


@pytest.mark.parametrize("first", range(0, N_SEEDS, BATCH))
def test_random_coincident_surfaces(first):
    """Equal t values on purpose (shapes under one transform, shared faces): ids, colours, lists in the reference's order."""
    failures = {}
    for seed in range(first, min(first + BATCH, N_SEEDS)):
        problems = compare(seed, depth=4, builder=scenes.random_tie_scene)
        world, _ = scenes.random_tie_scene(seed, 16, 16)
        rng = np.random.default_rng(seed)
        n = 256
        o = np.stack([rng.integers(-6, 7, size=n) * 1.0, rng.integers(-1, 4, size=n) * 0.5, np.full(n, -20.)], axis=1)
        d = np.tile(np.array([0., 0., 1.]), (n, 1))
        o[n // 2:] = o[n // 2:, [2, 1, 0]]   # the other half along +x
        d[n // 2:] = (1., 0., 0.)
        o[n // 2:, 0] = -20.
        gc, gt, gi = world.intersect_world_batch(o, d, 64)
        oc, ot, oi = ob.intersect_world_batch(world, o, d, 64)
        same_t = (gt == ot) | (np.isnan(gt) & np.isnan(ot))
        if not (np.array_equal(gc, oc) and same_t.all() and np.array_equal(gi, oi)):
            problems.append(f"lists: {int((gc != oc).sum())} counts, {int((~same_t).sum())} t, {int((gi != oi).sum())} ids differ")
        if problems:
            failures[seed] = problems
    assert not failures, failures


@pytest.mark.parametrize("first", range(0, N_SEEDS, BATCH))
def test_random_meshes(first):
    failures = {}
    for seed in range(first, min(first + BATCH, N_SEEDS)):
        problems = compare(seed, depth=3 if seed % 2 else 5, builder=scenes.random_mesh_scene)
        if problems:
            failures[seed] = problems
    assert not failures, failures


@pytest.mark.parametrize("first", range(0, N_SEEDS, BATCH))
def test_random_worlds_of_many_objects(first):
    """Worlds large enough for the top-level tree (64+ bounded objects): random primitives / CSG / groups between unbounded
    shapes, and scattered instances of one small mesh (flat or smooth) — the per-ray tree walk, the wavefront-uniform
    object rounds and the per-lane mesh instances against the oracle; every other seed also with each object left to the
    wavefront (RT_AMD_TLAS=2) and with ranges only (RT_AMD_TLAS=0)."""
    failures = {}
    for seed in range(first, min(first + BATCH, N_SEEDS)):
        mode = seed % 4
        env = {1: "2", 3: "0"}.get(mode)
        if env:
            os.environ["RT_AMD_TLAS"] = env
        try:
            if seed % 2:
                builder = lambda s, w, h: scenes.instanced_small_meshes(s, w, h, n=70 + (s * 7) % 60, smooth=bool((s // 2) % 2))
            else:
                builder = lambda s, w, h: scenes.many_objects_scene(s, w, h, n=80 + (s * 11) % 90)
            problems = compare(seed, depth=3 if seed % 3 else 5, builder=builder)
        finally:
            os.environ.pop("RT_AMD_TLAS", None)
        if problems:
            failures[seed] = problems
    assert not failures, failures


def _random_rays(rng, n):
    o = rng.normal(size=(n, 3))
    o = o / np.linalg.norm(o, axis=1)[:, None] * 13. * rng.uniform(0.02, 1.0, size=(n, 1)) + np.array([0., 1., 0.])
    target = rng.normal(size=(n, 3)) * 2.5 + np.array([0., 1., 0.])
    d = target - o
    d /= np.linalg.norm(d, axis=1)[:, None]
    # some axis-parallel rays (exact zeros in the direction: the slab tests' 0 * inf cases)
    k = n // 8
    axis = rng.integers(0, 3, size=k)
    d[:k] = 0.
    d[np.arange(k), axis] = rng.choice([-1., 1.], size=k)
    return o, d


@pytest.mark.parametrize("first", range(0, N_SEEDS, BATCH))
def test_random_lists_shadows_and_rays(first):
    """World::intersect_world lists (t and object id of every entry, in order), World::is_shadowed and
    World::color_at for arbitrary rays — origins inside and outside the objects — on the same random worlds."""
    failures = {}
    for seed in range(first, min(first + BATCH, N_SEEDS)):
        world, _ = (scenes.random_mesh_scene if seed % 2 else scenes.random_scene)(seed, 16, 16)
        o, d = _random_rays(np.random.default_rng(seed), 768)
        problems = []
        gc, gt, gi = world.intersect_world_batch(o, d, 64)
        oc, ot, oi = ob.intersect_world_batch(world, o, d, 64)
        same_t = (gt == ot) | (np.isnan(gt) & np.isnan(ot))
        if not (np.array_equal(gc, oc) and same_t.all() and np.array_equal(gi, oi)):
            problems.append(f"lists: {int((gc != oc).sum())} counts, {int((~same_t).sum())} t, {int((gi != oi).sum())} ids differ")
        if not np.array_equal(world.is_shadowed_batch(o), ob.is_shadowed_batch(world, o)):
            problems.append("is_shadowed differs")
        g_rgb, g_id, g_fl = world.color_at_batch(o[:256], d[:256], 3)
        o_rgb, o_id, o_fl = ob.color_at_batch(world, o[:256], d[:256], 3)
        err = np.abs(g_rgb - o_rgb)
        err[np.isnan(g_rgb) & np.isnan(o_rgb)] = 0.0
        err[np.isinf(g_rgb) & (g_rgb == o_rgb)] = 0.0
        if not (np.array_equal(g_id, o_id) and np.array_equal(g_fl, o_fl)):
            problems.append(f"color_at: {int((g_id != o_id).sum())} ids, {int((g_fl != o_fl).sum())} flags differ")
        if np.isnan(err).any() or not float(err.max()) < COLOR_TOL:
            problems.append(f"color_at: colour error {float(np.nanmax(err)):.3e}, {int(np.isnan(err).sum())} one-sided NaN")
        if problems:
            failures[seed] = problems
    assert not failures, failures


def _same_image(a, b, shift_ids=False):
    ids_b = b.hit_id
    if shift_ids and (a.hit_id > 0).any() and (b.hit_id > 0).any():  # object ids come from a process-wide counter
        shift = int(b.hit_id[b.hit_id > 0].min()) - int(a.hit_id[a.hit_id > 0].min())
        ids_b = np.where(b.hit_id > 0, b.hit_id - shift, 0)
    return np.array_equal(a.rgb, b.rgb, equal_nan=True) and np.array_equal(a.hit_id, ids_b) and np.array_equal(a.flags, b.flags)


@pytest.mark.parametrize("seed", range(6))
def test_random_worlds_on_every_render_path(seed, monkeypatch):
    """One random world at a frame size large enough for every path: the level-by-level walk (with
    rt_stats), the whole-frame-enqueue / graph-replayed path twice, a second camera on the same scene (new graph, queues that
    may have to grow), three tile shards, and the scene rebuilt without cull ranges and with the GPU-built BVH all
    give the same frame — and that frame matches the oracle on a sparse pixel sample."""
    import math
    builder = scenes.random_mesh_scene if seed % 2 else scenes.random_scene
    size, depth, stride = (1100, 640), 4, 23
    world, cam = builder(seed, *size)
    a = cam.render(world, depth)
    for _ in range(3):   # plain enqueue, graph capture, graph replay
        assert _same_image(a, cam.render(world, depth, want_stats=False))
    cam2 = rt.Camera(size[0], size[1], 0.9)
    cam2.transform = rt.view_transform((9. * math.cos(seed), 4., 9. * math.sin(seed)), (0., 0.5, 0.), (0., 1., 0.))
    b_fast = cam2.render(world, depth, want_stats=False)
    assert _same_image(cam2.render(world, depth), b_fast)
    assert _same_image(a, cam.render(world, depth, want_stats=False))  # back to the first view
    acc, ids = np.zeros_like(a.rgb), np.zeros_like(a.hit_id)
    for r in range(3):
        s = cam.render(world, depth, rank=r, n_ranks=3, want_stats=False)
        acc += np.nan_to_num(s.rgb, nan=-1e300)
        ids += s.hit_id
    assert np.array_equal(acc, np.nan_to_num(a.rgb, nan=-1e300)) and np.array_equal(ids, a.hit_id)
    o = ob.render(world, cam, depth, stride=stride)
    ys, xs = np.mgrid[0:size[1] - 1:stride, 0:size[0] - 1:stride]
    assert not (a.flags & rt.RT_FLAG_LIST_OVERFLOW).any()
    assert np.array_equal(a.hit_id[ys, xs], o.hit_id[ys, xs])
    assert np.array_equal(a.flags[ys, xs], o.flags[ys, xs])
    err = np.abs(a.rgb[ys, xs] - o.rgb[ys, xs])
    err[np.isnan(a.rgb[ys, xs]) & np.isnan(o.rgb[ys, xs])] = 0.
    assert not np.isnan(err).any() and float(err.max()) < COLOR_TOL
    for name, value in (("RT_AMD_CULL", "0"), ("RT_AMD_DEVICE_BVH", "1")):
        monkeypatch.setenv(name, value)
        world2, cam_again = builder(seed, *size)
        assert _same_image(a, cam_again.render(world2, depth, want_stats=False), shift_ids=True), name
        monkeypatch.delenv(name)


def test_random_canvases_to_ppm():
    """Canvas::canvas_to_ppm on the device against the host mirror, byte for byte: random widths around the 70-column
    wrap, values far outside [0, 1], infinities, NaN, exact halves, denormals."""
    import torch
    world = rt.World.default()
    rng = np.random.default_rng(77)
    specials = np.array([np.inf, -np.inf, np.nan, 1e300, -1e300, 5e-324, -0., 0.5 / 255., 254.5 / 255., 1., 1. + 2e-16, 0.999999])
    for _ in range(40):
        shape = (int(rng.integers(1, 40)), int(rng.integers(1, 70)))
        rgb = rng.uniform(-0.5, 1.5, size=shape + (3,))
        mask = rng.uniform(size=rgb.shape) < 0.15
        rgb[mask] = rng.choice(specials, size=int(mask.sum()))
        t = torch.from_numpy(np.ascontiguousarray(rgb)).to("cuda:0")
        torch.cuda.synchronize()
        text = rt.canvas_to_ppm_device(world.scene(), t.data_ptr(), shape[1], shape[0])
        assert text == rt.Image(rgb, None, None, None).to_ppm().encode(), shape
