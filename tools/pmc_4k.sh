#!/bin/bash
# Issue and lane utilisation of the tile kernel on the 4K / 16 spp workload (two rocprofv3 counter passes).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for pass in 1 2; do
  d=gpurun_out/pmc_4k/$pass; rm -rf "$d"
  if [ $pass = 1 ]; then C="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU";
  else C="SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_LDS"; fi
  timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d "$d" -- python3 bench.py --no-cpu-baseline --no-extra-passes --no-extra-workloads --steps 1 --warmup 1 --warmup-ms 0 --frames-per-step 8 --frames-in-flight ${FIF:-1} --workload bunny_4k_16spp > /dev/null 2>&1 || { echo "pass $pass failed"; exit 1; }
  python3 tools/pmc_table.py --kernel k_render_tile 4k=$d
done
