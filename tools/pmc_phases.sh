#!/bin/bash
# Instruction mix of the tile kernel with phases cut off (one rocprofv3 counter pass per variant).
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
C="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY"
for v in "full" "no phase 3" "no phase 2,3" "fill only"; do
    d="gpurun_out/pmc_${v// /_}"
    rm -rf "$d"
    timeout -s KILL 90 rocprofv3 --pmc $C --output-format csv -d "$d" -- python3 tools/phase_timing.py --only "$v" --rounds 1 > /dev/null 2>&1
done
(echo "8x8 tiles (a draw on its own):"
python3 tools/pmc_table.py full=gpurun_out/pmc_full "no3=gpurun_out/pmc_no_phase_3" "no23=gpurun_out/pmc_no_phase_2,3" "fill=gpurun_out/pmc_fill_only") | tee gpurun_out/pmc_phases.txt
# the same with 16x16 blocks (flags 67 = Lambert + invert + INV_BLOCKS), the way batched frames are traced
for v in "full" "no phase 3" "no phase 2,3"; do
    d="gpurun_out/pmc_blocks_${v// /_}"
    rm -rf "$d"
    timeout -s KILL 90 rocprofv3 --pmc $C --output-format csv -d "$d" -- python3 tools/phase_timing.py --only "$v" --rounds 1 --flags 67 > /dev/null 2>&1
done
(echo "16x16 blocks (batched frames):"
python3 tools/pmc_table.py full=gpurun_out/pmc_blocks_full "no3=gpurun_out/pmc_blocks_no_phase_3" "no23=gpurun_out/pmc_blocks_no_phase_2,3") | tee -a gpurun_out/pmc_phases.txt
