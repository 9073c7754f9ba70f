"""Per-kernel averages of every counter found in rocprofv3 --pmc output directories.

    python tools/pmc_table.py LABEL=DIR [LABEL=DIR ...] [--kernel k_render_tile]
"""
import collections
import csv
import glob
import os
import sys

from summarize_profiles import short


def main():
    kernel = "k_render_tile"
    runs = []
    args = sys.argv[1:]
    while args:
        a = args.pop(0)
        if a == "--kernel":
            kernel = args.pop(0)
        else:
            runs.append(a.split("=", 1))
    for label, d in runs:
        agg = collections.defaultdict(list)
        for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if short(r["Kernel_Name"]) == kernel:
                    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        print("%-16s " % label + "  ".join("%s=%.0f" % (k, sum(v) / len(v)) for k, v in sorted(agg.items())))


if __name__ == "__main__":
    main()
