#!/bin/bash
# Times bench workloads with each library under renderer_amd/variants/ (see tools/build_variants.sh).
#   tools/ab_variants.sh "<workload> <bench flags>" ["<workload> <flags>" ...]
O=gpurun_out/ab; mkdir -p $O
specs=("$@")
for lib in renderer_amd/variants/libinvert_*.so; do
  name=$(basename $lib .so); name=${name#libinvert_}
  for spec in "${specs[@]}"; do
    wl=${spec%% *}; flags=${spec#* }; [ "$flags" = "$spec" ] && flags=""
    INVERT_HIP_LIB=$PWD/$lib timeout -k 10 120 python3 bench.py --workload $wl --no-extra-workloads --no-cpu-baseline $flags > $O/$name.$wl.json 2> $O/$name.$wl.err || { echo "$name $wl FAILED"; tail -3 $O/$name.$wl.err; exit 1; }
    python3 -c "
import json
d=json.load(open('$O/$name.$wl.json')); print('%-10s %-28s %9.2f us/frame' % ('$name', '$wl', d['config']['us_per_frame']))"
  done
done
