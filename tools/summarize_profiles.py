"""Copy the rocprofv3 summaries worth keeping from gpurun_out/ into profiles/ (tracked).

    python tools/summarize_profiles.py TAG STATS_DIR [--fetch DIR] [--write DIR] [--workload NAME] [--note TEXT]

Writes profiles/TAG_kernel_stats.csv (the --kernel-trace --stats table) and, when the two PMC passes are given,
profiles/TAG_pmc.json with per-kernel averages of FETCH_SIZE and WRITE_SIZE (separate passes: the two do not fit
one TCC pass) and the HBM bytes per launch derived from them:

    hbm_bytes = 2 * FETCH_SIZE_KB * 1024 + WRITE_SIZE_KB * 1024

(on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced read; WRITE_SIZE is exact for 16-byte stores;
 other access widths are uncalibrated, so the read side is an estimate).
"""
import argparse
import collections
import csv
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = name.replace("(anonymous namespace)::", "").split("(")[0]
    name = name.split("<")[0]  # template arguments
    for prefix in ("void ", "inv::"):
        if name.startswith(prefix):
            name = name[len(prefix):]
    return name.split("::")[-1]


def pmc_avg(d, counter):
    # the newest file only: gpurun merges results into gpurun_out/, where an earlier pass of the same name may linger
    files = sorted(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)
    agg = collections.defaultdict(list)
    for f in files[-1:]:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("stats_dir")
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--workload", default="bunny_1080p_primary_lambert_invert")
    ap.add_argument("--note", default="")
    ap.add_argument("--frames-per-launch", type=int, default=1, help="frames one launch of the render kernel draws")
    a = ap.parse_args()
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    stats = sorted(glob.glob(os.path.join(a.stats_dir, "**", "*_kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    if stats:
        shutil.copy(stats[-1], os.path.join(out, a.tag + "_kernel_stats.csv"))
        print("copied", stats[-1])
    if a.fetch and a.write:
        fe, wr = pmc_avg(a.fetch, "FETCH_SIZE"), pmc_avg(a.write, "WRITE_SIZE")
        kernels = {}
        for k in sorted(set(fe) | set(wr)):
            f_kb, nf = fe.get(k, (0.0, 0))
            w_kb, nw = wr.get(k, (0.0, 0))
            kernels[k] = {"FETCH_SIZE_KB_avg": f_kb, "WRITE_SIZE_KB_avg": w_kb, "launches": max(nf, nw),
                          "hbm_bytes_per_launch": int(2 * f_kb * 1024 + w_kb * 1024)}
        doc = {"workload": a.workload, "note": a.note, "frames_per_launch": a.frames_per_launch,
               "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs; KB units; "
                         "read side doubled (gfx950 FETCH_SIZE counts 64 B per 128-B request)",
               "kernels": kernels}
        with open(os.path.join(out, a.tag + "_pmc.json"), "w") as f:
            json.dump(doc, f, indent=1)
        print(json.dumps({k: v for k, v in kernels.items() if "render" in k}, indent=1))


if __name__ == "__main__":
    main()
