import sys, math, time, json
sys.path.insert(0,'ray-tracer_amd'); sys.path.insert(0,'tests')
import numpy as np
import rt_amd as rt, oracle_bridge as ob, scenes
print("devices", rt.device_count(), flush=True)
def cmp(name, w, cam, depth=5):
    t=time.time(); g = cam.render(w, depth); tg=time.time()-t
    t=time.time(); g2 = cam.render(w, depth); tg2=time.time()-t
    t=time.time(); o = ob.render(w, cam, depth); to=time.time()-t
    ids = int((g.hit_id != o.hit_id).sum())
    err = float(np.nanmax(np.abs(g.rgb-o.rgb)))
    q = int((np.abs(g.quantised()-o.quantised())>1).sum())
    print(name, "gpu %.3fs/%.3fs oracle %.3fs id_mismatch %d max_err %.3e q>1 %d flags %s rays gpu %d orc %d ms_total %.3f trace %.3f shade %.3f" % (tg,tg2,to,ids,err,q,np.unique(g.flags),g2.stats.total_rays,o.stats.total_rays,g2.stats.ms_total,g2.stats.ms_trace,g2.stats.ms_queue), flush=True)
    return g,o
w = rt.World.default(); cam = rt.Camera(11,11,math.pi/2); cam.transform = rt.view_transform((0,0,-5),(0,0,0),(0,1,0))
g,o = cmp("default11", w, cam); print(g.pixel_at(5,5))
cmp("book", *scenes.book_world(320,160))
cmp("mixed", *scenes.mixed_scene(256,144))
cmp("cover", *scenes.cover_like_scene(120,120))
cmp("teapot96", *scenes.teapot_scene(96,54))
cmp("smooth", *scenes.smooth_teapot_alone(64,36))
cmp("teapotlow", *scenes.teapot_low(64,36))
w,c = scenes.teapot_scene(1920,1080)
for i in range(3):
    t=time.time(); g=c.render(w); print("C3 full", time.time()-t, g.stats.ms_total, g.stats.ms_trace, g.stats.ms_queue, g.stats.total_rays, g.stats.total_rays/g.stats.ms_total/1e3, "Mrays/s", flush=True)
