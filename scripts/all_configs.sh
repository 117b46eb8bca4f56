set -e
mkdir -p gpurun_out
timeout -k 10 200 python bench.py --scene teapot_low --no-cpu-baseline --steps 10 > gpurun_out/cfg_c2.json 2> gpurun_out/cfg_c2.err
timeout -k 10 200 python bench.py --scene teapot_smooth --no-cpu-baseline --steps 10 > gpurun_out/cfg_c3s.json 2> gpurun_out/cfg_c3s.err
timeout -k 10 200 python bench.py --scene mixed --width 2560 --height 1440 --no-cpu-baseline --steps 10 > gpurun_out/cfg_c4.json 2> gpurun_out/cfg_c4.err
timeout -k 10 300 python bench.py --scene grid --width 7680 --height 4320 --depth 8 --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/cfg_c5.json 2> gpurun_out/cfg_c5.err
python - <<'P'
import json
for c in ("c2","c3s","c4","c5"):
    d=json.load(open(f"gpurun_out/cfg_{c}.json"))
    print(c, d["value"], d["ms_per_step"], d["rays_per_frame"]["total"], {k:v["ms_per_frame"] for k,v in d["kernels"].items() if isinstance(v,dict)})
P
