set -e
for v in 1024 512 256; do
  for ch in 1 2 4; do
    RT_AMD_VERBOSE=1 RT_AMD_CHAINS=$ch RT_AMD_LIB=$PWD/ray-tracer_amd/lib/var_$v.so timeout -k 10 120 python bench.py --no-cpu-baseline --steps 30 --warmup 5 2> gpurun_out/ab_$v_$ch.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tblock $v chains $ch', d['ms_per_step'], d['value'], d['kernels']['k_trace']['ms_per_frame'], d['kernels']['k_secondary']['ms_per_frame'])"
  done
done
