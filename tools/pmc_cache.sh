#!/bin/bash
# L2 hit rate and VALU utilisation of the render kernels (two rocprofv3 counter passes per workload).
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for wl in bunny_1080p_primary_lambert_invert bunny_1080p_ao64 bunny_1080p_ao64_closeup bunny_4k_16spp; do
    for pass in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"; do
        d="gpurun_out/pmc_cache/${wl}_$(echo $pass | cut -d' ' -f1)"
        rm -rf "$d"
        timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d "$d" -- python3 bench.py --no-cpu-baseline --no-extra-passes --steps 1 --warmup 1 --warmup-ms 0 --frames-per-step $([ $wl = bunny_1080p_ao64_closeup ] && echo 4 || echo 24) --workload $wl > /dev/null 2>&1
    done
done
for wl in bunny_1080p_primary_lambert_invert bunny_1080p_ao64 bunny_1080p_ao64_closeup bunny_4k_16spp; do
    for k in k_render_tile_batch k_render_tile k_ao; do
        python3 tools/pmc_table.py --kernel $k "$wl/$k/L2=gpurun_out/pmc_cache/${wl}_TCC_HIT_sum" "$wl/$k/SQ=gpurun_out/pmc_cache/${wl}_SQ_BUSY_CU_CYCLES"
    done
done | grep "=" | tee gpurun_out/pmc_cache.txt
