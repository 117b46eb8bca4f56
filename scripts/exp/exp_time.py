"""Experiment (GPU, RT_EXP_TIME build): where the wavefront-cycles of the traversal kernels go."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ray-tracer_amd"))
import numpy as np
import rt_amd as rt
import scenes
w, cam = scenes.teapot_scene(1920, 1080)
prev = None
for depth in range(6):
    img = cam.render(w, depth, want_counters=True)
    s = img.stats
    cur = dict(wt=np.array(s.work_trace[:]), ws=np.array(s.work_shadow[:]), wc=np.array(s.work_container[:]))
    row = cur if prev is None else {k: cur[k] - prev[k] for k in cur}
    for nm in ("wt", "ws", "wc"):
        v = row[nm].astype(float)
        feed, load, trav, bvh, epi, total, setup, waves, chunks, lastfeed = v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], v[9]
        if waves == 0: continue
        print(f"level {depth} {nm}: waves {int(waves)} chunks {int(chunks)} | per wave kcycles: total {total/waves/1e3:7.1f} setup {setup/waves/1e3:6.1f} "
              f"final-feed-wait {lastfeed/waves/1e3:6.1f} | per chunk kcycles: feed {feed/max(chunks,1)/1e3:6.2f} load {load/max(chunks,1)/1e3:6.2f} "
              f"traverse {trav/max(chunks,1)/1e3:6.2f} (bvh {bvh/max(chunks,1)/1e3:6.2f}) epilogue {epi/max(chunks,1)/1e3:6.2f} | share of wave time: "
              f"[epi parts/chunk: nextfeed {v[10]/max(chunks,1)/1e3:.2f} recload {v[11]/max(chunks,1)/1e3:.2f} normal {v[12]/max(chunks,1)/1e3:.2f}] "
              f"setup {setup/total:.2f} feed {(feed+lastfeed)/total:.2f} load {load/total:.2f} items {(trav-bvh)/total:.2f} bvh {bvh/total:.2f} epi {epi/total:.2f}")
    prev = cur
