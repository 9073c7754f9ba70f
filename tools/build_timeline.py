"""Timeline of the last scene build in a rocprofv3 kernel trace of tools/config4.py.

    python tools/build_timeline.py gpurun_out/c4prof/vm/NNN_kernel_trace.csv
"""
import csv
import re
import sys


def short(name):
    m = re.search(r"\b(k_\w+)", name)  # this library's kernels; others (runtime copies and fills) by their first word
    return m.group(1) if m else re.split(r"[(<]", name.replace("void ", ""))[0].split("::")[-1] or name[:40]


rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "k_assemble" in r["Kernel_Name"]]
s = starts[-1]
t0 = int(rows[s]["Start_Timestamp"])
groups = {}
for r in rows[s:]:
    n = short(r["Kernel_Name"])
    if n.startswith(("k_render", "k_scatter", "k_vis_fill", "k_bin")):  # the first draw: the build is over
        break
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if "-v" in sys.argv:
        print("%8.1f %8.1f  %s" % ((a - t0) / 1e3, (b - a) / 1e3, n))
    g = groups.setdefault(n, [0, 0.0])
    g[0] += 1
    g[1] += (b - a) / 1e3
    end = b
print("last scene build of the run, on the GPU: %.1f us from the first kernel's start to the last kernel's end" % ((end - t0) / 1e3))
for n, (c, us) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
    print("%-28s %3d launches %8.1f us" % (n, c, us))
