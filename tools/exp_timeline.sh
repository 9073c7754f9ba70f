#!/bin/bash
# Kernel timeline of the 5 M-triangle scene build for each variant of the library named (see tools/exp_build.sh;
# "base" is the regular build): gpurun_out/exp/NAME.txt
cd "$(dirname "$0")/.."
R=$(pwd); O=$R/gpurun_out/exp; mkdir -p "$O"; export TMPDIR=/tmp
for name in "$@"; do
  if [ "$name" = base ]; then unset INVERT_HIP_LIB; else export INVERT_HIP_LIB=$R/renderer_amd/libinvert_hip_exp_$name.so; fi
  rm -rf "$O/trace_$name"
  (cd /tmp && timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d "$O/trace_$name" -- python3 $R/tools/config4.py --builds 8 > "$O/run_$name.txt" 2>&1) || { echo "$name failed"; tail -5 "$O/run_$name.txt"; exit 1; }
  python3 tools/build_timeline.py "$O"/trace_$name/*/*_kernel_trace.csv -v > "$O/$name.txt" 2>&1
  rm -rf "$O/trace_$name"
  echo "== $name"; grep -E "hit pixels|median" "$O/run_$name.txt" | cut -c1-150; cat "$O/$name.txt"
done
