#!/bin/bash
# Kernel timeline of the last of eight builds of a grid of bunnies (default 32 x 32: the 1024-instance scene), for each
# library given (default: the built one):
#   [SIDE=1] tools/trace_build.sh [path/to/libinvert_x.so ...]   ->  gpurun_out/bt/timeline_<name>.txt
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$R/gpurun_out/bt"
export TMPDIR=/tmp
for lib in "${@:-$R/renderer_amd/libinvert_hip.so}"; do
  lib=$(realpath "$lib"); name=$(basename "$lib" .so)
  export INVERT_HIP_LIB="$lib"
  (cd /tmp && timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/bt/trace_$name" -- python3 "$R/tools/config4.py" --builds 8 --side "${SIDE:-32}" > /dev/null 2>&1) || exit 1
  python3 "$R/tools/build_timeline.py" "$R"/gpurun_out/bt/trace_$name/*/*_kernel_trace.csv -v > "$R/gpurun_out/bt/timeline_$name.txt" 2>&1
  rm -rf "$R/gpurun_out/bt/trace_$name"
  echo "== $name"; grep -A6 "last scene build" "$R/gpurun_out/bt/timeline_$name.txt"
done
