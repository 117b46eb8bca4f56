"""bench.py prints ONE JSON line with the keys the driver's contract names (a small frame, no CPU baseline leg)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_keys():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-c5",
                          "--width", "480", "--height", "270"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None and d["unit"] == "Mrays/s"
    assert "workload" in d["config"] and d["value"] > 0 and d["ms_per_step"] > 0
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert abs(r["frac_vs_fma_spec"] - r["achieved"] / 78.6) < 1e-4
    h = d["roofline_hbm"]      # the HBM leg is measured traffic / launch time, or null — never algorithmic bytes / time
    # default --traffic auto: two rocprofv3 --pmc passes over a short child run of this workload, before the timed region
    if r["traffic"] is not None:
        assert r["traffic"] > 0 and r["traffic_source"].startswith("measured in this run") and h["traffic"] == r["traffic"]
        assert abs(h["achieved"] - r["traffic"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 0.02 * h["achieved"] and 0 < h["frac"] <= 1
    else:                      # (no rocprofv3 on this box: then there is no figure for this frame size)
        assert h["traffic"] is None and h["achieved"] is None and h["frac"] is None
    assert h["algorithmic_bytes_per_launch"] > 0
    # --traffic file: only the committed PMC files of the profiled workloads; this frame size has none: null, not the headline's figure
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-c5",
                          "--width", "480", "--height", "270", "--traffic", "file"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    f = json.loads([l for l in out.stdout.splitlines() if l.strip()][0])
    assert f["roofline"]["traffic"] is None and f["roofline_hbm"]["traffic"] is None and f["roofline_hbm"]["frac"] is None
    host = d["rt_render_host"]  # SURVEY 8(d)'s wording: frames complete in host memory; the pipelined pair beside the blocking call
    assert 0 < host["ms_per_frame"] and 0 < host["ms_per_frame_blocking_rt_render"] <= host["ms_per_frame_pageable_buffer"] * 1.5
    assert d["sharded"] is None


@pytest.mark.gpu
def test_sharded_bench_path_prints_one_json_line_too():
    """The multi-rank code path (RCCL gather + rt_assemble), rehearsed with one rank: RCCL prints a version banner on
    stdout when its communicator is created, and stdout must still carry the JSON line alone."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rehearse-sharded", "--steps", "3", "--warmup", "1",
                          "--no-cpu-baseline", "--no-c5", "--width", "480", "--height", "270"], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["value"] > 0 and d["scaling"] == "strong"
    sh = d["sharded"]          # the keys the first real N > 1 run explains itself with
    assert sh["ranks_seen"] == 1 and sh["gather"] == "f64" and sh["bytes_per_pixel_on_the_wire"] == 24 and sh["backend"] == "nccl"
    assert len(sh["per_rank"]) == 1
    for k in ("rank", "shard_ms", "gather_ms", "assemble_ms"):
        assert k in sh["per_rank"][0], k
    assert sh["per_rank"][0]["shard_ms"] > 0 and sh["per_rank"][0]["gather_ms"] >= 0 and sh["per_rank"][0]["assemble_ms"] > 0
    # the timed region renders this rank's shards three in flight (rt_render_shard_begin); what the gather + scatter leave on rank 0
    # is bit for bit the frame one whole-frame render gives
    assert d["frames_in_flight"] == 3 and d["one_frame_at_a_time"]["ms_per_frame"] > 0
    assert sh["assembled_frame_equals_whole_frame_render"] is True


@pytest.mark.gpu
def test_sharded_bench_path_with_the_u8_gather():
    """--gather u8: scale_color on the shard (rt_quantise), 3 bytes per pixel through the gather, rt_assemble_quantised."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29534")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rehearse-sharded", "--gather", "u8", "--steps", "3", "--warmup", "1",
                          "--no-cpu-baseline", "--no-c5", "--width", "480", "--height", "270"], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["sharded"]["gather"] == "u8" and d["sharded"]["bytes_per_pixel_on_the_wire"] == 3 and d["value"] > 0
    assert d["sharded"]["assembled_frame_equals_whole_frame_render"] is True
