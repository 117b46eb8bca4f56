"""usage: python scripts/kernel_resources.py [extra hipcc flags]   (CPU box)
VGPRs, spilled VGPRs, scratch bytes and occupancy of every traversal / shading kernel variant, from the compiler's own
resource-usage remarks (device-only compile of csrc/rt_api.hip; nothing is linked or run)."""
import re
import subprocess
import sys

cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-std=c++17", "-c", "--cuda-device-only",
       "-Rpass-analysis=kernel-resource-usage", *sys.argv[1:], "ray-tracer_amd/csrc/rt_api.hip", "-o", "/tmp/rt_api_dev.o"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
name, rec = None, {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = m.group(1)
        rec[name] = {}
        continue
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("spill", r"VGPRs Spill: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("sgpr", r"TotalSGPRs: (\d+)")):
        m2 = re.search(pat, line)
        if m2 and name:
            rec[name][key] = m2.group(1)
rows = []
for n, r in rec.items():
    d = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    d = re.sub(r"\(.*", "", d).replace("void rt::", "")
    if any(k in d for k in ("k_trace", "k_walk", "k_list", "k_shadow_points", "k_comps", "k_shade")):
        rows.append("%-48s VGPR %4s spill %4s scratch %6s SGPR %4s occ %s" % (d, r.get('vgpr'), r.get('spill'), r.get('scratch'), r.get('sgpr'), r.get('occ')))
print("\n".join(sorted(rows)))
