set -e
B="python3 bench.py --no-cpu-baseline --steps 200 --warmup 40"
for v in 0 1; do
  INV_BINS=$v timeout -k 10 120 $B > gpurun_out/ab_default_$v.json
  INV_BINS=$v timeout -k 10 120 $B --workload bunny_4k_16spp > gpurun_out/ab_4k_$v.json
  INV_BINS=$v timeout -k 10 120 $B --workload bunny_1080p_ao64 > gpurun_out/ab_ao_$v.json
  INV_BINS=$v timeout -k 10 120 $B --workload bunny_x1024_1080p --steps 5 --warmup 1 > gpurun_out/ab_x1024_$v.json
  INV_BINS=$v timeout -k 10 120 $B --frames-in-flight 1 --frames-per-launch 1 > gpurun_out/ab_single_$v.json
done
echo done
