#!/bin/bash
# Counters of the one-lane-per-triangle kernels on the 5 M-triangle frame (two rocprofv3 counter passes).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for pass in 1 2 3; do
  d=gpurun_out/pmc_x1024/$pass; rm -rf "$d"
  if [ $pass = 1 ]; then C="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU";
  elif [ $pass = 2 ]; then C="SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_BUSY_CYCLES";
  else C="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum"; fi
  timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d "$d" -- python3 bench.py --no-cpu-baseline --no-extra-passes --no-extra-workloads --steps 1 --warmup 1 --warmup-ms 0 --frames-per-step 4 --frames-in-flight 1 --workload bunny_x1024_1080p > /dev/null 2>&1 || { echo "pass $pass failed"; exit 1; }
  for k in k_scatter_triangles k_scatter_resolve4; do python3 tools/pmc_table.py --kernel $k $k=$d; done
done
