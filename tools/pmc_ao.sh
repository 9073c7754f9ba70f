#!/bin/bash
# Instruction mix and lane utilisation of the occlusion kernel (one rocprofv3 counter pass).
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
d=gpurun_out/pmc_ao
rm -rf "$d"
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU \
    --output-format csv -d "$d" -- python3 bench.py --no-cpu-baseline --no-extra-passes --steps 1 --warmup 1 --frames-per-step 8 --workload bunny_1080p_ao64 > /dev/null 2>&1
python3 tools/pmc_table.py --kernel k_ao ao=$d | tee gpurun_out/pmc_ao.txt
