#!/bin/bash
# The config-4 driver (tools/config4.py: every draw path of the 5 M-triangle scene) under each library of
# renderer_amd/variants/ (see tools/build_variants.sh).
cd "$(dirname "$0")/.."
for lib in renderer_amd/variants/libinvert_*.so; do
  echo "== $(basename $lib)"
  INVERT_HIP_LIB=$PWD/$lib timeout -k 10 200 python3 tools/config4.py "$@" 2>&1 | grep -E "ms per frame" | cut -c1-110 || exit 1
done
