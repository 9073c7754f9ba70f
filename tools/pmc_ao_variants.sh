#!/bin/bash
# Counters of the occlusion kernel for every library under renderer_amd/variants/ (close-up workload).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for lib in renderer_amd/variants/libinvert_*.so; do
  name=$(basename $lib .so); name=${name#libinvert_}
  for pass in 1 2; do
    d=gpurun_out/pmc_aov/$name.$pass; rm -rf "$d"
    if [ $pass = 1 ]; then C="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU";
    else C="SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_SCA"; fi
    INVERT_HIP_LIB=$PWD/$lib timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d "$d" -- python3 bench.py --no-cpu-baseline --no-extra-passes --no-extra-workloads --steps 1 --warmup 1 --warmup-ms 0 --frames-per-step 4 --frames-in-flight 1 --workload bunny_1080p_ao64_closeup > /dev/null 2>&1 || { echo "$name pass $pass failed"; exit 1; }
    python3 tools/pmc_table.py --kernel k_ao $name=$d
  done
done
