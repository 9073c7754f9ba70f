#!/bin/bash
# Counters of the scene build's kernels on the 5 M-triangle scene (rocprofv3 counter passes, one run each).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for pass in ${PASSES:-1 2 3}; do
  d=gpurun_out/pmc_build/$pass; rm -rf "$d"
  if [ $pass = 1 ]; then C="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU";
  elif [ $pass = 2 ]; then C="SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_BUSY_CYCLES";
  elif [ $pass = 4 ]; then C="TA_BUSY_avr TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE GRBM_TA_BUSY SQ_WAVES SQ_BUSY_CYCLES";
  elif [ $pass = 5 ]; then C="SPI_RA_REQ_NO_ALLOC_CSN SPI_RA_WAVE_SIMD_FULL_CSN SPI_RA_VGPR_SIMD_FULL_CSN SPI_RA_LDS_CU_FULL_CSN GRBM_GUI_ACTIVE GRBM_SPI_BUSY SQ_WAVES SQ_LEVEL_WAVES";
  else C="SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC"; fi
  timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d "$d" -- python3 tools/config4.py --builds 3 > /dev/null 2>&1 || { echo "pass $pass failed"; exit 1; }
  for k in ${KERNELS:-k_build_chunks k_build_level k_sort_scatter}; do python3 tools/pmc_table.py --kernel $k $k=$d; done
done
