#!/bin/bash
# Instruction fetch of the tile kernel when a draw runs on its own (every launch starts with cold instruction caches):
# fetches, their summed time in flight (SQ_IFETCH_LEVEL / SQ_IFETCH = cycles per fetch), hits and misses of the
# instruction caches.  Two rocprofv3 counter passes.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for pass in 1 2; do
  d=gpurun_out/pmc_ifetch/$pass; rm -rf "$d"
  if [ $pass = 1 ]; then C="SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY";
  else C="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_BUSY_CYCLES GRBM_GUI_ACTIVE"; fi
  timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d "$d" -- python3 bench.py --no-cpu-baseline --no-extra-passes --no-extra-workloads --steps 1 --warmup 1 --warmup-ms 0 --frames-per-step 8 --frames-in-flight 1 --frames-per-launch 1 ${WORKLOAD:+--workload $WORKLOAD} > /dev/null 2>&1 || { echo "pass $pass failed"; exit 1; }
  python3 tools/pmc_table.py --kernel ${KERNEL:-k_render_tile} draw=$d
done
