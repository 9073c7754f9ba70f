#!/bin/bash
# One pass of every measurement kept under profiles/: bench lines, rocprofv3 kernel stats, the two HBM counter
# passes, phase timing, per-tile statistics and the instruction-mix table.  Run on the GPU box from the repo root:
#     bash tools/profile_round.sh TAG
set -e
TAG=${1:-r03}
R=$(pwd)
O=$R/gpurun_out/$TAG
rm -rf "$O" && mkdir -p "$O"
export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-extra-passes --steps 60 --warmup 6 --warmup-ms 0"
timeout -k 10 200 python3 bench.py > "$O/bench_default.json"
timeout -k 10 200 python3 bench.py --workload bunny_4k_16spp > "$O/bench_4k_16spp.json"
timeout -k 10 200 python3 bench.py --workload bunny_1080p_ao64 > "$O/bench_ao64.json"
timeout -k 10 250 python3 bench.py --workload bunny_x1024_1080p --steps 5 --warmup 1 > "$O/bench_x1024.json"
timeout -k 10 250 python3 bench.py --workload bunny_1080p_ao64_closeup --no-cpu-baseline --steps 4 --warmup 1 --frames-per-step 16 > "$O/bench_ao64_closeup.json"
timeout -k 10 200 python3 bench.py --no-cpu-baseline --frames-in-flight 1 --frames-per-launch 1 > "$O/bench_single_draw.json"
(cd /tmp && timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- $B > /dev/null 2>&1)
(cd /tmp && timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/fetch" -- $B > /dev/null 2>&1)
(cd /tmp && timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/write" -- $B > /dev/null 2>&1)
(cd /tmp && timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_ao" -- $B --workload bunny_1080p_ao64 > /dev/null 2>&1)
timeout -k 10 120 python3 tools/phase_timing.py > "$O/phase_timing.txt" 2>&1
timeout -k 10 120 python3 tools/phase_timing.py --frames-in-flight 3 > "$O/phase_timing_3_in_flight.txt" 2>&1
timeout -k 10 120 tools/micro/host_loop "$(python3 -c 'import sys; sys.path.insert(0, "tests"); import objfmt; print(objfmt.bunny_obj_path())')" > "$O/host_loop.txt" 2>&1
timeout -k 10 120 python3 tools/tile_stats.py > "$O/tile_stats.txt" 2>&1
timeout -k 10 120 python3 tools/tile_stats.py --blocks > "$O/block_stats.txt" 2>&1
timeout -k 10 120 python3 tools/phase_timing.py --flags 67 > "$O/phase_timing_blocks.txt" 2>&1
bash tools/pmc_phases.sh > /dev/null 2>&1 && cp gpurun_out/pmc_phases.txt "$O/pmc_phases.txt"
timeout -k 10 300 python3 tools/config4.py > "$O/config4.txt" 2>&1
(cd /tmp && timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d "$O/build_trace" -- python3 $R/tools/config4.py --builds 8 > /dev/null 2>&1)
python3 tools/build_timeline.py "$O"/build_trace/*/*_kernel_trace.csv -v > "$O/build_timeline.txt" 2>&1
bash tools/pmc_cache.sh > /dev/null 2>&1 && cp gpurun_out/pmc_cache.txt "$O/pmc_cache.txt"
bash tools/pmc_ao.sh > /dev/null 2>&1 && cp gpurun_out/pmc_ao.txt "$O/pmc_ao.txt"
bash tools/pmc_4k.sh > "$O/pmc_4k.txt" 2>&1
bash tools/pmc_x1024.sh > "$O/pmc_x1024.txt" 2>&1
timeout -k 10 200 python3 tools/build_probe.py > "$O/build_probe.txt" 2>&1
KERNELS="k_build_chunks k_build_level k_build_cut k_sort_scatter k_sort_hist k_sort_runs k_morton" bash tools/pmc_build.sh > "$O/pmc_build.txt" 2>&1
# the 5 M-triangle frame and its scene builds: kernel statistics and HBM bytes (FETCH_SIZE, WRITE_SIZE: a pass each)
BX="python3 $R/bench.py --no-cpu-baseline --no-extra-passes --steps 2 --warmup 1 --warmup-ms 0 --frames-per-step 8 --workload bunny_x1024_1080p"
(cd /tmp && timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_x1024" -- $BX > /dev/null 2>&1)
(cd /tmp && timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/fetch_x1024" -- $BX > /dev/null 2>&1)
(cd /tmp && timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/write_x1024" -- $BX > /dev/null 2>&1)
bash tools/pmc_read_hit.sh > /dev/null 2>&1 && cp gpurun_out/pmc_read_hit.txt "$O/pmc_read_hit.txt"
timeout -k 10 250 python3 bench.py --workload bunny_x13924_1080p --no-cpu-baseline --steps 3 --warmup 1 --frames-per-step 8 > "$O/bench_x13924.json"
timeout -k 10 250 python3 tools/config4.py --side 118 --builds 5 > "$O/config4_118.txt" 2>&1
(cd /tmp && timeout -s KILL 250 rocprofv3 --kernel-trace --output-format csv -d "$O/build_trace_118" -- python3 $R/tools/config4.py --side 118 --builds 3 > /dev/null 2>&1)
python3 tools/build_timeline.py "$O"/build_trace_118/*/*_kernel_trace.csv -v > "$O/build_timeline_118.txt" 2>&1
timeout -k 10 120 python3 tools/memset_probe.py > "$O/memset_probe.txt" 2>&1
INVERT_HIP_LIB=$R/renderer_amd/libinvert_hip_timing.so timeout -k 10 200 python3 tools/fill_probe.py timing-build > "$O/fill_probe.txt" 2>&1
echo done
