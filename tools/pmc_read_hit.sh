#!/bin/bash
# Read hit rate of the XCDs' L2s, stores left out: TCC_READ (read requests the L2s received) against TCC_EA0_RDREQ (the
# reads they passed on to the fabric), per kernel, one rocprofv3 counter pass per workload.
#   bash tools/pmc_read_hit.sh  ->  gpurun_out/pmc_read_hit.txt
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=gpurun_out/pmc_read_hit.txt; : > $out
run() { # workload, extra bench arguments, kernels
  wl=$1; extra=$2; shift 2
  d=gpurun_out/pmc_read_hit/$wl; rm -rf "$d"
  timeout -s KILL 150 rocprofv3 --pmc TCC_READ_sum TCC_EA0_RDREQ_sum TCC_WRITE_sum TCC_EA0_WRREQ_sum --output-format csv -d "$d" -- python3 bench.py --no-cpu-baseline --no-extra-passes --steps 1 --warmup 1 --warmup-ms 0 --workload $wl $extra > /dev/null 2>&1 || { echo "$wl: pass failed" >> $out; return; }
  for k in "$@"; do python3 tools/pmc_table.py --kernel $k "$wl/$k=$d" >> $out; done
}
run bunny_1080p_primary_lambert_invert "--frames-per-step 24" k_render_tile_batch
run bunny_1080p_primary_lambert_invert "--frames-per-step 8 --frames-in-flight 1 --frames-per-launch 1" k_render_tile
run bunny_x1024_1080p "--frames-per-step 4 --frames-in-flight 1" k_scatter_triangles k_scatter_resolve4 k_build_chunks k_sort_scatter
run bunny_1080p_ao64 "--frames-per-step 8" k_ao
python3 - <<'PY' >> gpurun_out/pmc_read_hit.txt
import re
for line in open("gpurun_out/pmc_read_hit.txt").read().splitlines():
    v = {k: float(x) for k, x in re.findall(r"(\w+)=([0-9.]+)", line)}
    if v.get("TCC_READ_sum"):
        print("%-60s read hit rate %.3f (%.0f read requests, %.0f passed on); writes %.0f, passed on %.0f" % (
            line.split()[0], 1.0 - v["TCC_EA0_RDREQ_sum"] / v["TCC_READ_sum"], v["TCC_READ_sum"], v["TCC_EA0_RDREQ_sum"],
            v.get("TCC_WRITE_sum", 0), v.get("TCC_EA0_WRREQ_sum", 0)))
PY
cat $out
