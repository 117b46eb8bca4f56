#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X-native render path.

Workload (BASELINE.json metric: "Mrays/sec + ms/frame, teapot.obj 1920x1080 depth-5"): configs[2] — teapot.obj as the
reference parser loads it (6320 flat triangles) + glass sphere + mirror sphere + checkered reflective floor, one
light, 1920x1080, depth 5 (tests/scenes.py:teapot_scene).  One "step" = one frame = Camera::render over the whole
(hsize-1)x(vsize-1) pixel domain.  Scene tables are resident in HBM before the timed region; each step ends with the
frame assembled in device memory on rank 0 (N = 1: rt_render_frame writes the row-major frame directly; N > 1:
shards gathered with RCCL over xGMI, then scattered by rt_assemble).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  value = rays traced by all ranks per frame / max-over-ranks wall time per frame.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

# Three frames in flight are rendered on three streams beside the caller's and the scene's own: the HIP runtime maps a process's
# streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share a queue run one after the other, so the
# third render lane needs more queues than the default (include/rt_abi.h: rt_scene_opts.render_lanes).  Read by the runtime
# when it starts: set before torch / the library are loaded; inherited by the rocprofv3 child runs.
_q = os.environ.get("GPU_MAX_HW_QUEUES", "")
if os.environ.get("RT_BENCH_HW_QUEUES") or not (_q.isdigit() and int(_q) >= 6):   # (a setting that is already large enough stays)
    os.environ["GPU_MAX_HW_QUEUES"] = os.environ.get("RT_BENCH_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "ray-tracer_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

# algorithmic flops per unit of work, counted from the reference formulas as written
# (1 per f64 add/sub/mul/div/sqrt; compares, selects, abs, negation not counted).
F_XFORM = 64          # Ray::transform: two 4x4 * tuple products (src/ray.rs:25-30)
F_BVH_BOX = 12        # acceleration box test: 6 sub + 6 mul (reciprocal direction), min/max not counted
F_GATE = 18           # Bound::intersects (src/bound.rs:85-116)
F_TRI_FULL, F_TRI_DET, F_TRI_U, F_TRI_V = 55, 16, 29, 47   # Triangle::intersect exit paths (src/triangle.rs:62-87)
F_PRIM = (38, 2, 18, 30, 36)                                # sphere, plane, cube, cylinder wall, cone wall
F_PREPARE = 130       # prepare_computations with one parent level (60 + 70 per level)
F_LIGHTING = 75       # Material::lighting per light (+ one powf)
F_REFLECT, F_REFRACT, F_SCHLICK = 12, 25, 20
F_RAYGEN = 80         # Camera::ray_for_pixel (src/camera.rs:37-60): two matrix * tuple products and a normalize
B_BVH_NODE, B_MESH_TRI = 128, 80   # bytes fetched per BVH node / per triangle tested (rt_program.h)
# bytes of level state read + written per ray by each stage (rt_device.h: Level columns touched)
# (k_trace: a primary ray is read from its columns, 48; a ray of levels 1.. is made from its tag and its parent's hit,
# 4 + 76, and stored, 48, when it hits something)
B_RAY_IO = {"k_trace": 48 + 17, "k_trace_spawned": 4 + 76 + 17, "k_trace_hit": 56, "k_trace_spawned_hit": 48, "shadow": 4 + 48 + 1,
            "container": 4 + 64 + 16 + 4, "k_queue": 4 + 8 + 8, "k_shade": 4 + 4 + 72 + 16 + 1 + 24 + 24}

FP64_FMA_SPEC_TFLOPS = 78.6   # AMD's FP64 vector figure for MI355X: counts an FMA as two operations
FP64_VALU_PEAK_TFLOPS = 39.3  # MI355X FP64 vector: 78.6 TFLOP/s counting FMA as 2 (AMD spec); with -ffp-contract=off
                              # (mandatory for reference parity) every op is a separate mul/add => half of that
FP64_VALU_MEASURED_TFLOPS = 35.6  # scripts/fp64_roof.hip on the box (profiles/fp64_roof.json): v_mul_f64 + v_add_f64 chains, 8 waves/SIMD
HBM_PEAK_GBS = 8000.0         # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable)


def work_flops(w):
    full = w[3] - w[4] - w[5] - w[6]
    f = w[0] * F_XFORM + w[1] * F_BVH_BOX + w[2] * F_GATE
    f += full * F_TRI_FULL + w[4] * F_TRI_DET + w[5] * F_TRI_U + w[6] * F_TRI_V
    f += sum(w[7 + k] * F_PRIM[k] for k in range(5))
    return f


def work_bytes(w):
    return w[12] * B_BVH_NODE + w[3] * B_MESH_TRI


def _cpu_band(job):
    """Worker of the all-cores CPU baseline: one row band of the strided sample, rendered by the C oracle."""
    scene, width, height, depth, stride, y0, y1, grid = job
    import oracle_bridge as ob
    import scenes
    builders = {"teapot": scenes.teapot_scene, "teapot_low": scenes.teapot_low, "mixed": scenes.mixed_scene,
                "book": scenes.book_world, "grid": lambda wd, ht: scenes.instanced_grid(wd, ht, grid=grid),
                "teapot_smooth": lambda wd, ht: scenes.teapot_scene(wd, ht, obj_text=scenes.smooth_teapot_obj())}
    w, cam = builders[scene](width, height)
    t0 = time.perf_counter()
    img = ob.render(w, cam, depth, stride=stride, rows=(y0, y1))
    return img.stats.total_rays, img.stats.rays_primary, time.perf_counter() - t0


def cpu_baseline_all_cores(args):
    """The oracle on every host core this job may use (the GPU box gives a one-GPU job 16), over the same strided
    sample as the single-core figure, cut into row bands.  Runs in forked workers BEFORE this process touches the
    GPU (a process that has initialised HIP must not fork/exec helpers on this pool)."""
    import multiprocessing as mp
    cores = max(1, min(16, os.cpu_count() or 1))
    rows = args.height - 1
    bands = cores * 4   # 4 bands per worker: glass / teapot rows cost far more than sky rows
    edges = [rows * b // bands for b in range(bands + 1)]
    jobs = [(args.scene, args.width, args.height, args.depth, args.cpu_stride, edges[b], edges[b + 1], args.grid) for b in range(bands)]
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(cores) as pool:
        res = pool.map(_cpu_band, jobs, chunksize=1)
    wall = time.perf_counter() - t0
    rays = sum(r[0] for r in res)
    return {"value": round(rays / wall / 1e6, 6), "unit": "Mrays/s", "cores": cores, "rays": rays, "wall_s": round(wall, 2),
            "cpu_s": round(sum(r[2] for r in res), 2)}


def measure_traffic(args):
    """HBM bytes per launch of every kernel of this workload, from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: one
    counter per pass, as MI355X_MICROARCH.md prescribes) of a two-frame child run of this script; units and the gfx950
    half-count correction of FETCH_SIZE as in scripts/pmc_traffic.py.  None when it cannot be done here (no rocprofv3,
    this run is being profiled itself, a pass failed or ran into its time limit): the caller falls back to profiles/."""
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    profiled = any(k.startswith(("ROCPROF", "ROCP_", "ROCTX")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
    if not os.path.exists(exe) or (profiled and args.traffic != "measure"):
        return None
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    try:
        import pmc_traffic
    except Exception:
        return None
    child = ["python3", os.path.abspath(__file__), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-c5", "--traffic", "file", "--in-flight", "1",
             "--scene", args.scene, "--grid", str(args.grid), "--width", str(args.width), "--height", str(args.height), "--depth", str(args.depth)]
    tmp = tempfile.mkdtemp(prefix="rt_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    per = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            # a pass over two frames takes 10-20 s (8K: 30 s); a profiler that hangs is killed, and the committed figure is used
            cmd = ["timeout", "-s", "KILL", "90", exe, "--pmc", counter, "--output-format", "csv", "-d", out, "--", *child]
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=120)
            if r.returncode != 0:
                return None
            per[counter] = pmc_traffic.per_kernel(out, counter)
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    res = {}
    for k in set(per["FETCH_SIZE"]) | set(per["WRITE_SIZE"]):
        fk, fn = per["FETCH_SIZE"].get(k, [0.0, 0])
        wk, wn = per["WRITE_SIZE"].get(k, [0.0, 0])
        if k.startswith("k_") and fn and wn:
            res[k] = {"hbm_read_bytes_per_launch": int(fk * 1024.0 * 2.0 / fn), "hbm_write_bytes_per_launch": int(wk * 1024.0 / wn),
                      "hbm_bytes_per_launch": int(fk * 1024.0 * 2.0 / fn + wk * 1024.0 / wn), "launches_sampled": fn}
    return res or None


def main():
    # stdout carries exactly ONE line, the JSON result: libraries that print there (RCCL's version banner at communicator
    # creation, for one) are sent to stderr by pointing file descriptor 1 at it for the rest of the run
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)  # 0.33 s of frames: long enough for the driver's GPU-activity sampler
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--depth", type=int, default=5)
    ap.add_argument("--scene", default="teapot", choices=["teapot", "teapot_smooth", "teapot_low", "mixed", "book", "grid"])
    ap.add_argument("--grid", type=int, default=4, help="--scene grid: grid x grid teapots (13 = 1.07 M triangles, SURVEY 8(d) C5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--in-flight", type=int, default=3, choices=[1, 2, 3],
                    help="frames (N > 1: this rank's shards) in flight in the timed region.  3 (default) or 2: rt_render_frame_begin / "
                         "rt_render_shard_begin / rt_render_end, the frames rendered side by side on streams of their own (a frame's "
                         "launch tails are filled by the other frames' launches); 1: one blocking rt_render_frame / rt_render_shard per "
                         "frame (what the profile scripts trace: launches not overlapped)")
    ap.add_argument("--rehearse-sharded", action="store_true",
                    help="N = 1 only: take the N > 1 code path (rt_render_shard, RCCL gather in a one-rank group, rt_assemble_strided)")
    ap.add_argument("--no-c5", action="store_true", help="skip the C5 workloads (grid 4 and grid 13 at 8K) reported beside the headline")
    ap.add_argument("--gather", default="f64", choices=["f64", "u8"],
                    help="sharded path: what travels over xGMI. f64: the f64 RGB shards (24 B/px); u8: scale_color on each rank's "
                         "shard first (rt_quantise), 3 B/px, the root assembles the u8 frame canvas_to_ppm reads")
    ap.add_argument("--cpu-stride", type=int, default=12)
    ap.add_argument("--traffic", default="auto", choices=["auto", "measure", "file"],
                    help="roofline.traffic (HBM bytes per launch of the dominant kernel). measure: two rocprofv3 --pmc passes "
                         "(FETCH_SIZE, WRITE_SIZE; each its own short child run of this script) before this process touches "
                         "the GPU; file: the committed figure under profiles/; auto: measure at N = 1 when rocprofv3 is there "
                         "and this run is not itself being profiled, else file")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
        args.gpus = world

    measured_traffic = None
    if world == 1 and not args.rehearse_sharded and args.traffic != "file":
        measured_traffic = measure_traffic(args)   # child processes; before torch / HIP are initialised in this one
    cpu_all = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        cpu_all = cpu_baseline_all_cores(args)   # before torch / HIP are initialised in this process

    import numpy as np
    import torch
    import torch.distributed as dist
    import rt_amd as rt
    import scenes

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU implementation")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    sharded = world > 1 or args.rehearse_sharded   # the multi-rank code path (also with one rank, as a rehearsal)
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    builders = {"teapot": scenes.teapot_scene, "teapot_low": scenes.teapot_low, "mixed": scenes.mixed_scene,
                "book": scenes.book_world, "grid": lambda wd, ht: scenes.instanced_grid(wd, ht, grid=args.grid),
                # north_star's literal target: the same scene with SmoothTriangles (generated vertex normals; no bundled
                # OBJ yields any, F2).  Next to other shapes the reference panics on some pixels (F5b): those are flagged
                "teapot_smooth": lambda wd, ht: scenes.teapot_scene(wd, ht, obj_text=scenes.smooth_teapot_obj())}

    def make_job(w, cam, depth, width, height):
        """The per-frame step of one workload on this rank: scene resident, buffers allocated, nothing timed yet."""
        scene = w.scene(local_rank)          # flatten + upload + BVH build: outside the timed region
        cd = cam.desc()
        counts = [int(rt.lib.rt_shard_pixel_count(C.byref(cd), r, world)) for r in range(world)]
        n_max = max(counts)
        # N > 1: two shard buffers used in turn, so that the RCCL gather of frame k (on torch's stream) overlaps the render
        # of frame k+1 (on the library's stream) with no host wait in between; a buffer is rendered into again only after
        # the gather that read it two frames earlier has finished (an event recorded behind that gather — by then long done)
        # (six with shards in flight: up to three being rendered, one being gathered, two free)
        NB, D = 6, max(args.in_flight, 1)
        shards = [torch.zeros((n_max, 3), dtype=torch.float64, device=dev) for _ in range(NB if sharded else 0)]
        gather_done = [None] * NB
        frame = torch.zeros((height, width, 3), dtype=torch.float64, device=dev) if rank == 0 else None
        frame_b = [torch.zeros((height, width, 3), dtype=torch.float64, device=dev) for _ in range(2 if not sharded else 0)]  # frames in flight
        # the root gathers into one [world, n_max, 3] buffer: rt_assemble_strided reads the padded shards where they land
        gathered_all = torch.zeros((world, n_max, 3), dtype=torch.float64, device=dev) if (rank == 0 and sharded) else None
        gathered = list(gathered_all.unbind(0)) if gathered_all is not None else None
        u8 = sharded and args.gather == "u8"
        if u8:   # scale_color per rank before the gather (src/canvas.rs:86-98): 8x fewer bytes over xGMI; the root keeps a u8 frame
            shards_q = [torch.zeros((n_max, 3), dtype=torch.uint8, device=dev) for _ in range(NB)]
            gathered_all = torch.zeros((world, n_max, 3), dtype=torch.uint8, device=dev) if rank == 0 else None
            gathered = list(gathered_all.unbind(0)) if gathered_all is not None else None
            frame = torch.zeros((height, width, 3), dtype=torch.uint8, device=dev) if rank == 0 else None
        opts = rt.RenderOpts(max_depth=depth, rank=rank, n_ranks=world, want_counters=0, stream=None)
        torch_stream = torch.cuda.current_stream(dev).cuda_stream if sharded else None
        state = {"k": 0}

        def step(stats=None, phases=None):
            """phases: a dict that receives this step's shard_ms (host clock around rt_render_shard, which returns when the
            shard is complete in HBM), gather_ms and assemble_ms (events on torch's stream, the one those two are enqueued on)"""
            if not sharded:
                # one GPU: Camera::render straight into the row-major frame in HBM (rt_render_frame), one blocking call
                rc = rt.lib.rt_render_frame(scene, C.byref(cd), C.byref(opts), frame.data_ptr(), None, None,
                                            C.byref(stats) if stats is not None else None)
                if rc < 0:
                    raise RuntimeError(rt.last_error())
                return
            b = state["k"] & 1
            state["k"] += 1
            if gather_done[b] is not None:
                gather_done[b].synchronize()          # the gather that read this buffer two frames ago
            # returns when the shard is complete in HBM (the library waits once per frame for its ray counters)
            t_host = time.perf_counter()
            rc = rt.lib.rt_render_shard(scene, C.byref(cd), C.byref(opts), shards[b].data_ptr(), None, None,
                                        C.byref(stats) if stats is not None else None)
            if rc < 0:
                raise RuntimeError(rt.last_error())
            t_host = time.perf_counter() - t_host
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)] if phases is not None else None
            if e:
                e[0].record()
            send = shards[b]
            if u8:   # on torch's stream, like the gather behind it
                rc = rt.lib.rt_quantise(scene, shards[b].data_ptr(), n_max * 3, shards_q[b].data_ptr(), C.c_void_p(torch_stream))
                if rc < 0:
                    raise RuntimeError(rt.last_error())
                send = shards_q[b]
            dist.gather(send, gathered, dst=0)        # RCCL gather over xGMI: no reduction, peers -> root; enqueued, not waited for
            if e:
                e[1].record()
            if rank == 0:                             # stream-ordered behind the gather on torch's stream
                if u8:
                    rc = rt.lib.rt_assemble_quantised(scene, C.byref(cd), world, gathered_all.data_ptr(), n_max, frame.data_ptr(),
                                                      C.c_void_p(torch_stream))
                else:
                    rc = rt.lib.rt_assemble_strided(scene, C.byref(cd), world, gathered_all.data_ptr(), None, n_max, frame.data_ptr(), None,
                                                    C.c_void_p(torch_stream))
                if rc < 0:
                    raise RuntimeError(rt.last_error())
            ev = torch.cuda.Event(enable_timing=phases is not None)
            ev.record()
            gather_done[b] = ev
            if e:
                ev.synchronize()
                phases["shard_ms"] = phases.get("shard_ms", 0.) + t_host * 1e3
                phases["gather_ms"] = phases.get("gather_ms", 0.) + e[0].elapsed_time(e[1])
                phases["assemble_ms"] = phases.get("assemble_ms", 0.) + e[1].elapsed_time(ev)
                phases["frames"] = phases.get("frames", 0) + 1

        def frames_in_flight(n):
            """n frames, D in flight (one GPU): frame k is handed to rt_render_frame_begin as soon as frame k - D has been
            retired; every one of the n frames is complete in its device frame when this returns"""
            bufs = (frame.data_ptr(), frame_b[0].data_ptr(), frame_b[1].data_ptr())
            for k in range(n):
                if k >= D and rt.lib.rt_render_end(scene) < 0:
                    raise RuntimeError(rt.last_error())
                if rt.lib.rt_render_frame_begin(scene, C.byref(cd), C.byref(opts), bufs[k % D], None, None) < 0:
                    raise RuntimeError(rt.last_error())
            for _ in range(min(n, D)):
                if rt.lib.rt_render_end(scene) < 0:
                    raise RuntimeError(rt.last_error())

        comm_stream = torch.cuda.Stream(dev) if sharded else None   # non-blocking: the render slots do not wait for a gather

        def post(b):
            """shard buffer b is complete in HBM: [quantise,] gather to rank 0 and scatter into the frame, enqueued on the
            communication stream; the event behind them frees the buffer"""
            with torch.cuda.stream(comm_stream):
                cs = C.c_void_p(comm_stream.cuda_stream)
                send = shards[b]
                if u8:
                    if rt.lib.rt_quantise(scene, shards[b].data_ptr(), n_max * 3, shards_q[b].data_ptr(), cs) < 0:
                        raise RuntimeError(rt.last_error())
                    send = shards_q[b]
                dist.gather(send, gathered, dst=0)
                if rank == 0:
                    if u8:
                        rc = rt.lib.rt_assemble_quantised(scene, C.byref(cd), world, gathered_all.data_ptr(), n_max, frame.data_ptr(), cs)
                    else:
                        rc = rt.lib.rt_assemble_strided(scene, C.byref(cd), world, gathered_all.data_ptr(), None, n_max, frame.data_ptr(), None, cs)
                    if rc < 0:
                        raise RuntimeError(rt.last_error())
                ev = torch.cuda.Event()
                ev.record()
                gather_done[b] = ev

        def shards_in_flight(n):
            """n frames on the N > 1 path with this rank's shards D in flight (rt_render_shard_begin / rt_render_end, rendered
            side by side like whole frames at N = 1): shard k is begun once shard k - D has been retired and handed to
            its gather; the gather and scatter of a frame run on a stream of their own beside the renders.  Every rank takes
            the same steps in the same order, so the collectives match; all n frames are assembled on rank 0 once the
            device is idle (the caller's barrier + synchronize)."""
            for g in gather_done:
                if g is not None:
                    g.synchronize()
            for op, k, b in rt.shard_pipeline(n, D, NB):   # (the order of steps: rt_amd.shard_pipeline)
                if op == "end":
                    if rt.lib.rt_render_end(scene) < 0:
                        raise RuntimeError(rt.last_error())
                elif op == "begin":
                    if gather_done[b] is not None:
                        gather_done[b].synchronize()      # the gather that read this buffer six frames ago
                    if rt.lib.rt_render_shard_begin(scene, C.byref(cd), C.byref(opts), shards[b].data_ptr(), None, None) < 0:
                        raise RuntimeError(rt.last_error())
                else:
                    post(b)
            state["k"] = 0

        def verify():
            """untimed, every rank calls it: three more frames through the shards-in-flight path, then rank 0 renders the WHOLE
            frame by itself (rt_render_frame, rank 0 of 1) and compares it bit for bit with what the gather + scatter left
            there (u8 gather: with scale_color of it) — a pixel's value must not depend on the partition (src/camera.rs:65-71)"""
            if not sharded:
                return None
            shards_in_flight(3)
            barrier()
            if rank != 0:
                return None
            whole = torch.full((height, width, 3), -1., dtype=torch.float64, device=dev)
            torch.cuda.synchronize()
            o1 = rt.RenderOpts(max_depth=depth, rank=0, n_ranks=1, want_counters=0, stream=None)
            if rt.lib.rt_render_frame(scene, C.byref(cd), C.byref(o1), whole.data_ptr(), None, None, None) < 0:
                raise RuntimeError(rt.last_error())
            if u8:
                q = torch.empty((height, width, 3), dtype=torch.uint8, device=dev)
                if rt.lib.rt_quantise(scene, whole.data_ptr(), height * width * 3, q.data_ptr(), None) < 0:
                    raise RuntimeError(rt.last_error())
                return bool(torch.equal(q, frame))
            return bool(torch.equal(whole.view(torch.int64), frame.view(torch.int64)))

        return {"step": step, "scene": scene, "cd": cd, "opts": opts, "world_obj": w, "keep": (shards, frame, frame_b, gathered_all), "frame": frame,
                "frames_in_flight": shards_in_flight if sharded else frames_in_flight, "verify": verify}

    def sharded_report(step, frames=5):
        """Per-rank phase times of the sharded path, a few untimed frames each waited for: what the first real N > 1 run needs
        to explain itself (is it the shard render's launch floor, the gather, or the scatter?)."""
        if not sharded:
            return None
        ph = {}
        for _ in range(frames):
            step(None, ph)
        mine = torch.tensor([ph["shard_ms"] / frames, ph["gather_ms"] / frames, ph["assemble_ms"] / frames], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        return {"ranks_seen": dist.get_world_size(), "backend": dist.get_backend(), "gather": args.gather,
                "bytes_per_pixel_on_the_wire": 3 if args.gather == "u8" else 24,
                "per_rank": [{"rank": r, "shard_ms": round(float(t[0]), 4), "gather_ms": round(float(t[1]), 4),
                              "assemble_ms": round(float(t[2]), 4)} for r, t in enumerate(every)],
                "what": "means over %d untimed frames, each waited for: shard_ms = host clock around rt_render_shard (returns when the "
                        "shard is in HBM), gather_ms / assemble_ms = events on the stream the gather and the scatter are enqueued on "
                        "(assemble runs on rank 0 only)" % frames}

    def timed(step, steps, warmup, many=None):
        """W untimed steps, then exactly K steps between barrier + synchronize on both sides; max over ranks, in ms per step.
        `many(n)`: n steps in one call (two frames in flight) instead of n calls of step()."""
        if many:
            many(warmup)
        else:
            for _ in range(warmup):
                step()
        barrier()
        t0 = time.perf_counter()
        if many:
            many(steps)
        else:
            for _ in range(steps):
                step()
        barrier()
        tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        if sharded:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item()) / steps * 1e3

    def ray_counts(step):
        st = rt.Stats()
        step(st)
        local = torch.tensor([st.rays_primary, st.rays_reflect, st.rays_refract, st.rays_shadow, st.shaded_nodes],
                             dtype=torch.int64, device=dev)
        if sharded:
            dist.all_reduce(local)
        return st, [int(x) for x in local.tolist()]

    def barrier():
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()

    w, cam = builders[args.scene](args.width, args.height)
    job = make_job(w, cam, args.depth, args.width, args.height)
    step, scene, cd, opts = job["step"], job["scene"], job["cd"], job["opts"]
    piped = args.in_flight >= 2
    ms_per_step = timed(step, args.steps, args.warmup, job["frames_in_flight"] if piped else None)
    # one frame at a time (the latency of a frame; what `value` was until round 3's last step), beside the pipelined rate
    ms_single = timed(step, min(args.steps, 100), 3) if piped else ms_per_step

    # ---- SURVEY 8(d)'s literal metric, reported next to `value`: rt_render = the same frame with HOST output buffers
    # (one 49.8 MB D2H copy of the f64 frame at 1080p included).  Never `value`: that is the HBM-resident rate.
    host_ms = host_ms_pageable = host_ms_pipelined = None
    if world == 1 and not sharded:
        def time_rt_render(host_rgb):
            for _ in range(2):
                rt.lib.rt_render(scene, C.byref(cd), C.byref(opts), host_rgb.ctypes.data_as(rt.c_dp), None, None, None)
            n_host = max(3, min(args.steps, 10))
            t0 = time.perf_counter()
            for _ in range(n_host):
                rc = rt.lib.rt_render(scene, C.byref(cd), C.byref(opts), host_rgb.ctypes.data_as(rt.c_dp), None, None, None)
                if rc < 0:
                    raise RuntimeError(rt.last_error())
            return (time.perf_counter() - t0) / n_host * 1e3
        host_ms = time_rt_render(rt.pinned_array((args.height, args.width, 3)))      # buffer from rt_host_alloc
        host_ms_pageable = time_rt_render(np.zeros((args.height, args.width, 3)))     # any malloc'ed buffer

        host_bufs = [rt.pinned_array((args.height, args.width, 3)) for _ in range(6)]

        def time_rt_render_pipelined(depth):
            """rt_render_begin / rt_render_end: the same frames into page-locked host buffers used in turn, `depth` frames in
            flight (some being rendered side by side, the others being copied on copy streams beside the renders); every frame
            is complete in host memory when its end returns"""
            def run(n):
                for k in range(n):
                    if k >= depth and rt.lib.rt_render_end(scene) < 0:
                        raise RuntimeError(rt.last_error())
                    rc = rt.lib.rt_render_begin(scene, C.byref(cd), C.byref(opts), host_bufs[k % depth].ctypes.data_as(rt.c_dp), None, None)
                    if rc < 0:
                        raise RuntimeError(rt.last_error())
                for _ in range(min(n, depth)):
                    if rt.lib.rt_render_end(scene) < 0:
                        raise RuntimeError(rt.last_error())
            run(2 * depth)
            n_host = max(6, min(args.steps, 40))
            t0 = time.perf_counter()
            run(n_host)
            return (time.perf_counter() - t0) / n_host * 1e3
        # 4: what two render lanes could use; 6: three lanes.  --in-flight 1 (the profile scripts): one at a time here too, so that
        # every launch of the run lasts what it lasts alone and a kernel trace's averages agree with the HIP events'
        host_by_depth = {d: time_rt_render_pipelined(d) for d in ((4, 6) if piped else (1,))}
        host_depth = min(host_by_depth, key=host_by_depth.get)
        host_ms_pipelined = host_by_depth[host_depth]

    # ---- untimed: ray accounting, per-kernel HIP-event times, instrumented work counters (this rank's shard) ----
    st, rays = ray_counts(step)
    n_ev = 5
    ev = {"ms_trace": 0.0, "ms_walk": 0.0, "ms_queue": 0.0, "ms_kshade": 0.0, "ms_total": 0.0}
    for _ in range(n_ev):
        s2 = rt.Stats()
        step(s2)
        for k in ev:
            ev[k] += getattr(s2, k) / n_ev
    opts.want_counters = 1
    sc = rt.Stats()
    step(sc)
    opts.want_counters = 0
    total_rays = sum(rays[:4])
    sharded_info = sharded_report(step)
    same = job["verify"]()
    if sharded_info is not None:
        sharded_info["assembled_frame_equals_whole_frame_render"] = same

    # ---- second workload, for the scaling sweep (the driver runs this file at N = 1, 2, 4, 8 with default arguments):
    # a 1.8 ms frame cannot scale across GPUs (launch floors and the gather dominate), C5 can.  Reported beside the
    # headline, never as `value`: 4 x 4 instanced teapots (101,120 triangles), 7680x4320, depth 8, same tile sharding.
    c5 = c5_13 = None
    if args.scene == "teapot" and not args.no_c5:
        barrier()
        del job, step

        def run_c5(grid, what):
            w5, cam5 = scenes.instanced_grid(7680, 4320, grid=grid)
            job5 = make_job(w5, cam5, 8, 7680, 4320)
            ms5 = timed(job5["step"], 20, 4, job5["frames_in_flight"] if piped else None)
            ms5_single = timed(job5["step"], 10, 2) if piped else ms5
            _, rays5 = ray_counts(job5["step"])
            out5 = {"workload": what, "ms_per_step": round(ms5, 4), "value": round(sum(rays5[:4]) / (ms5 * 1e-3) / 1e6, 3), "unit": "Mrays/s",
                    "steps": 20, "warmup": 4, "rays_per_frame": sum(rays5[:4]), "n_gpus": world, "frames_in_flight": args.in_flight,
                    "one_frame_at_a_time": {"ms_per_frame": round(ms5_single, 4), "Mrays_s": round(sum(rays5[:4]) / (ms5_single * 1e-3) / 1e6, 3)},
                    "sharded": sharded_report(job5["step"], 3)}
            same5 = job5["verify"]()
            if out5["sharded"] is not None:
                out5["sharded"]["assembled_frame_equals_whole_frame_render"] = same5
            torch.cuda.synchronize()
            w5.invalidate()
            del job5
            return out5
        c5 = run_c5(4, "configs[4] geometry at grid 4: 16 instanced teapots (101,120 triangles) + floor, 7680x4320, depth 8")
        # the survey's scale (1-4 M triangles): 13 x 13 teapots + 13 x 13 triangles.obj = 1,068,418 triangles
        c5_13 = run_c5(13, "configs[4] at the survey's scale, grid 13: 169 instanced teapots + 169 triangles.obj (1,068,418 triangles) "
                           "+ floor, 7680x4320, depth 8")

    if rank == 0:
        n_lights = len(w.lights)
        # k_trace = one launch per level: the closest hits of level d AND the shadow rays of level d-1 (+ one launch for
        # the last level's shadow rays); k_walk = the n1/n2 container walks of a level.  The instrumented run counts
        # the three kinds of work apart (work_trace / work_shadow / work_container).
        w_tr = [a + b for a, b in zip(sc.work_trace, sc.work_shadow)]
        works = {"k_trace": w_tr, "k_walk": list(sc.work_container)}
        ms = {"k_trace": ev["ms_trace"], "k_walk": ev["ms_walk"], "k_queue": ev["ms_queue"], "k_shade": ev["ms_kshade"]}
        launches = {"k_trace": st.n_trace_launches, "k_walk": st.n_walk_launches,
                    "k_queue": max(st.n_trace_launches - 1, 1), "k_shade": max(st.n_trace_launches - 1, 1)}
        spawned = st.rays_reflect + st.rays_refract
        traced = st.rays_primary + spawned
        flops = {k: work_flops(v) for k, v in works.items()}
        flops["k_trace"] += st.shaded_nodes * F_PREPARE   # prepare_computations runs in k_trace's epilogue,
        flops["k_trace"] += st.rays_reflect * F_REFLECT + st.rays_refract * F_REFRACT   # the ray spawn where it fetches its rays
        if os.environ.get("RT_AMD_FUSE_RAYGEN", "1") != "0":   # ... and ray_for_pixel for level 0's (rt_scene_opts.fuse_raygen, the default)
            flops["k_trace"] += st.rays_primary * F_RAYGEN
        flops["k_queue"] = 0
        flops["k_shade"] = st.shaded_nodes * F_LIGHTING * n_lights + st.rays_refract * F_SCHLICK
        # (hits of levels 1.. are not counted apart: the spawned rays' share of the shaded nodes prices their stored rays)
        spawned_hits = st.shaded_nodes * spawned // max(traced, 1)
        fused_raygen = os.environ.get("RT_AMD_FUSE_RAYGEN", "1") != "0"
        # (primary rays made by the launch that traces them: no 48-byte ray to read, and the ray is stored only with a hit)
        primary_io = st.rays_primary * 17 + (st.shaded_nodes - spawned_hits) * 48 if fused_raygen else st.rays_primary * B_RAY_IO["k_trace"]
        nbytes = {"k_trace": work_bytes(works["k_trace"]) + primary_io + spawned * B_RAY_IO["k_trace_spawned"] +
                  st.shaded_nodes * B_RAY_IO["k_trace_hit"] + spawned_hits * B_RAY_IO["k_trace_spawned_hit"] + st.rays_shadow * B_RAY_IO["shadow"],
                  "k_walk": work_bytes(works["k_walk"]) + st.rays_container * B_RAY_IO["container"],
                  "k_queue": traced * B_RAY_IO["k_queue"], "k_shade": traced * B_RAY_IO["k_shade"]}
        dom = max(("k_trace", "k_walk"), key=lambda k: ms[k])
        n_launch = max(launches[dom], 1)
        pmc = None
        # PMC-measured HBM traffic of the profiled workloads (scripts/pmc_collect.sh, committed): the headline frame and
        # C5 at the survey's scale; any other workload has no measurement and reports null
        pmc_name = None
        if args.scene == "teapot" and (args.width, args.height, args.depth) == (1920, 1080, 5):
            pmc_name = "pmc_traffic.json"
        elif args.scene == "grid" and (args.grid, args.width, args.height, args.depth) == (13, 7680, 4320, 8):
            pmc_name = "r03_pmc_traffic_c5_grid13.json"
        pmc_path = os.path.join(ROOT, "profiles", pmc_name or "none")
        traffic_source = None
        if measured_traffic and dom in measured_traffic:
            pmc = measured_traffic[dom]["hbm_bytes_per_launch"]
            traffic_source = ("measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, one pass each over a two-frame child "
                              "run of this workload before the timed region (%d launches of %s sampled; read = FETCH_SIZE KiB x 1024 x 2, "
                              "the gfx950 half-count correction, %d B; written = WRITE_SIZE KiB x 1024, %d B)"
                              % (measured_traffic[dom]["launches_sampled"], dom, measured_traffic[dom]["hbm_read_bytes_per_launch"],
                                 measured_traffic[dom]["hbm_write_bytes_per_launch"]))
        elif os.path.exists(pmc_path):
            try:
                pmc = json.load(open(pmc_path)).get(dom, {}).get("hbm_bytes_per_launch")
            except Exception:
                pmc = None
            if pmc is not None:
                traffic_source = ("profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload, committed; not collected "
                                  "in this run)" % pmc_name)
        achieved = flops[dom] / (ms[dom] * 1e-3) / 1e12 if ms[dom] > 0 else 0.0
        roofline = {"kernel": dom, "bound": "fp64_valu", "achieved": round(achieved, 4), "peak": FP64_VALU_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(achieved / FP64_VALU_PEAK_TFLOPS, 5), "traffic": pmc,
                    "frac_vs_fma_spec": round(achieved / FP64_FMA_SPEC_TFLOPS, 5),
                    "peak_measured": FP64_VALU_MEASURED_TFLOPS, "frac_of_measured": round(achieved / FP64_VALU_MEASURED_TFLOPS, 5),
                    "launches_per_frame": n_launch, "avg_launch_ms": round(ms[dom] / n_launch, 5),
                    "source": "HIP events on the launch stream around every launch of %d rt_stats frames; those frames run "
                              "the launch schedule of a timed frame, one frame at a time and with no host wait between launches; "
                              "with two frames in flight the same launches overlap the other frame's and last longer each" % n_ev,
                    "flops_source": "work counters of one instrumented frame (want_counters=1) x SURVEY 8(d) constants",
                    "traffic_source": traffic_source,
                    "algorithmic_flops_per_launch": int(flops[dom] / n_launch),
                    "algorithmic_bytes_per_launch": int(nbytes[dom] / n_launch),
                    "note": "f64 VALU-bound path, no MFMA (no dense contraction); peak = AMD FP64 vector spec / 2 because "
                            "reference parity forbids FMA contraction"}
        # the whole frame at the rate `value` is quoted on (frames in flight fill each other's launch tails): every kernel's
        # algorithmic flops per frame over the wall time per frame
        frame_flops = float(sum(flops.values())) * (total_rays / max(sum(
            [st.rays_primary, st.rays_reflect, st.rays_refract, st.rays_shadow]), 1))   # (N > 1: rank 0's shard scaled to the frame)
        roofline["whole_frame"] = {
            "algorithmic_flops_per_frame": int(frame_flops),
            "at_value_rate": {"ms_per_frame": round(ms_per_step, 4), "achieved": round(frame_flops / (ms_per_step * 1e-3) / 1e12, 4),
                              "frac": round(frame_flops / (ms_per_step * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS, 5)},
            "one_frame_at_a_time": {"ms_per_frame": round(ms_single, 4), "achieved": round(frame_flops / (ms_single * 1e-3) / 1e12, 4),
                                    "frac": round(frame_flops / (ms_single * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS, 5)},
            "note": "all kernels' algorithmic flops of one frame / wall time per frame (launch gaps, queue compaction, shading and "
                    "the idle tails included): `frac` above is the dominant kernel alone, timed launch by launch with one frame "
                    "at a time"}
        # The HBM leg is priced on MEASURED traffic (PMC passes), never on algorithmic bytes: those count 128 B per BVH node
        # even when the node comes out of LDS and 80 B per triangle even when L2 serves it, so bytes / time can exceed what
        # HBM delivers.  No PMC file for this workload: null.
        launch_s = ms[dom] * 1e-3 / n_launch if ms[dom] > 0 else 0.0
        hbm_ach = pmc / launch_s / 1e9 if (pmc is not None and launch_s > 0) else None
        roofline_hbm = {"kernel": dom, "bound": "hbm", "achieved": None if hbm_ach is None else round(hbm_ach, 2), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": None if hbm_ach is None else round(min(hbm_ach / HBM_PEAK_GBS, 1.0), 5), "traffic": pmc,
                        "traffic_source": roofline["traffic_source"],
                        "algorithmic_bytes_per_launch": int(nbytes[dom] / n_launch),
                        "algorithmic_GBps": round(nbytes[dom] / (ms[dom] * 1e-3) / 1e9, 2) if ms[dom] > 0 else None,
                        "note": "achieved = measured HBM bytes per launch (rocprofv3 FETCH_SIZE + WRITE_SIZE passes) / average launch "
                                "duration; algorithmic_* = SURVEY 8(d) bytes, served largely by LDS and L2, reported apart"}
        kernels = {k: {"ms_per_frame": round(ms[k], 4), "launches_per_frame": launches[k], "algorithmic_flops": int(flops[k]),
                       "algorithmic_bytes": int(nbytes[k]),
                       "tflops": round(flops[k] / (ms[k] * 1e-3) / 1e12, 4) if ms[k] > 0 else None} for k in ms}
        kernels["device_ms_per_frame_rank0"] = round(ev["ms_total"], 4)
        # the reference's own algorithm tests every leaf for every ray (no culling at all: its triangle
        # groups have infinite boxes); reported so that the BVH's savings are not mistaken for roofline gains
        n_tri, n_other = int(w.flatten().contents.n_triangles), 3
        per_ray = n_tri * F_TRI_U + F_PRIM[1] + 2 * F_PRIM[0] + 3 * F_XFORM   # most tests leave at the u test
        reference_work = {"flops_per_ray_brute_force": per_ray, "flops_per_frame_brute_force": int(per_ray * total_rays),
                          "flops_per_frame_this_build": int(sum(flops.values())),
                          "note": "every ray x every leaf (%d triangles + %d other leaves), triangle tests priced at the "
                                  "u-reject exit" % (n_tri, n_other)} if args.scene == "teapot" else None

        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            import oracle_bridge as ob
            t0 = time.perf_counter()
            ref = ob.render(w, cam, args.depth, stride=args.cpu_stride)
            dt = time.perf_counter() - t0
            one = round(ref.stats.total_rays / dt / 1e6, 6)
            cpu = {"value": cpu_all["value"] if cpu_all else one, "unit": "Mrays/s", "cores": cpu_all["cores"] if cpu_all else 1,
                   "kind": "port", "single_core_value": one, "all_cores": cpu_all,
                   "sample": "every %dth pixel in x and y of the same frame (%d primary rays, %d rays in all; %.1f s on one "
                             "core, then the same sample cut into row bands over the job's host cores); C restatement of "
                             "the reference (oracle/rt_oracle.c), brute force over every triangle per ray like the "
                             "reference, -O2 -ffp-contract=off, the reference itself is single-threaded; the Rust crate "
                             "cannot be built here (no rustc/cargo)" % (args.cpu_stride, ref.stats.rays_primary,
                                                                        ref.stats.total_rays, dt)}
        out = {"metric": "Mrays/s", "value": round(total_rays / (ms_per_step * 1e-3) / 1e6, 3), "unit": "Mrays/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "configs[2]: teapot.obj (6320 flat triangles, reference parser) + glass & mirror "
                                      "spheres + checkered floor, 1 light, %dx%d, depth %d" % (args.width, args.height, args.depth)
                          if args.scene == "teapot" else args.scene,
                          "scene": args.scene, "resolution": [args.width, args.height], "depth": args.depth,
                          "parallelism": "tile-shard x%d, RCCL gather" % world if sharded else "single GPU"},
               "rays_per_frame": {"primary": rays[0], "reflect": rays[1], "refract": rays[2], "shadow": rays[3],
                                  "total": total_rays, "shaded_nodes": rays[4]},
               "value_is": ("rt_render_shard_begin / rt_render_end on every rank: Camera::render frame after frame, the frame cut into 8x8 "
                            "tiles dealt round-robin to the ranks, each rank's shards %s in flight (rendered side by side on streams of their own, "
                            "a shard's launch tails filled by the next frame's shard), every retired shard gathered to rank 0 (RCCL, "
                            "one gather per frame) and scattered into the row-major frame in HBM on a communication stream beside the "
                            "renders; all K frames are assembled on rank 0 inside the timed region (barrier + device synchronize on both "
                            "sides, max over ranks).  `one_frame_at_a_time`: one blocking rt_render_shard per frame and rank, then the "
                            "gather (what this path timed until round 3's last step)" % ("THREE" if args.in_flight == 3 else "TWO") if (piped and sharded) else
                            "rt_render_frame_begin / rt_render_end: Camera::render frame after frame with the Canvas left in HBM "
                            "(row-major f64), %s frames in flight — rendered side by side on streams of their own, a frame's launch "
                            "tails filled by the other frames' launches; all K frames are complete inside the timed region; scene resident "
                            "before it.  `one_frame_at_a_time` is the same frame with one blocking rt_render_frame per frame (a "
                            "frame's latency, and what `value` was up to round 3's r03b set); the roofline object is quoted on "
                            "launches of such frames" % ("THREE" if args.in_flight == 3 else "TWO") if piped else
                            "rt_render_frame: Camera::render with the Canvas left in HBM (row-major f64), one blocking call per frame, "
                            "scene resident before the timed region"),
               "frames_in_flight": args.in_flight,
               "one_frame_at_a_time": {"ms_per_frame": round(ms_single, 4), "Mrays_s": round(total_rays / (ms_single * 1e-3) / 1e6, 3)},
               "rt_render_host": None if host_ms is None else {
                   "ms_per_frame": round(host_ms_pipelined, 4), "Mrays_s": round(total_rays / (host_ms_pipelined * 1e-3) / 1e6, 3),
                   "frames_in_flight": host_depth, "ms_per_frame_by_frames_in_flight": {str(d): round(v, 4) for d, v in host_by_depth.items()},
                   "ms_per_frame_blocking_rt_render": round(host_ms, 4),
                   "ms_per_frame_pageable_buffer": round(host_ms_pageable, 4),
                   "what": "SURVEY 8(d)'s wording of the metric — every frame complete in a HOST f64 buffer, the D2H copy of %.1f MB "
                           "included; PCIe-inclusive, so never `value`.  ms_per_frame: rt_render_begin / rt_render_end, frame "
                           "after frame into page-locked buffers (rt_host_alloc) used in turn, timed with four and with six frames in "
                           "flight (up to three being rendered side by side, the others being copied on copy streams) and quoted for the "
                           "faster of the two (frames_in_flight); ms_per_frame_blocking_rt_render: one rt_render call per frame "
                           "(render, then copy) into a page-locked buffer; the last figure: rt_render into pageable memory"
                           % (args.height * args.width * 24 / 1e6)},
               "sharded": sharded_info,
               "c5_grid4_8k": c5, "c5_grid13_8k": c5_13, "roofline": roofline, "roofline_hbm": roofline_hbm, "kernels": kernels, "reference_algorithm": reference_work,
               "cpu_baseline": cpu}
        print(json.dumps(out), file=json_out, flush=True)
    if sharded:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
