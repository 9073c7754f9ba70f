"""Copy one pass of tools/profile_round.sh (gpurun_out/TAG) into profiles/ under the round's names.

    python tools/collect_profiles.py TAG [--prefix r01_final] [--frames-per-launch 4]
"""
import argparse
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--prefix", default="r03_final")
    ap.add_argument("--frames-per-launch", type=int, default=4)
    a = ap.parse_args()
    src, dst = os.path.join(ROOT, "gpurun_out", a.tag), os.path.join(ROOT, "profiles")
    names = {
        "bench_default.json": "bench_bunny_1080p.json", "bench_4k_16spp.json": "bench_bunny_4k_16spp.json",
        "bench_ao64.json": "bench_bunny_1080p_ao64.json", "bench_x1024.json": "bench_bunny_x1024_1080p.json", "phase_timing.txt": "tile_phase_timing.txt",
        "phase_timing_3_in_flight.txt": "tile_phase_timing_3_in_flight.txt", "host_loop.txt": "host_loop.txt",
        "tile_stats.txt": "tile_stats.txt", "block_stats.txt": "block_stats.txt",
        "phase_timing_blocks.txt": "block_phase_timing.txt", "pmc_phases.txt": "tile_instruction_mix.txt", "config4.txt": "config4.txt",
        "pmc_cache.txt": "l2_valu_counters.txt", "build_timeline.txt": "lbvh_build_timeline.txt", "pmc_ao.txt": "ao_instruction_mix.txt",
        "bench_ao64_closeup.json": "bench_bunny_1080p_ao64_closeup.json", "bench_single_draw.json": "bench_bunny_1080p_single_draw.json",
        "pmc_4k.txt": "4k_16spp_counters.txt", "pmc_read_hit.txt": "l2_read_hit_rate.txt", "bench_x13924.json": "bench_bunny_x13924_1080p.json",
        "config4_118.txt": "config4_69m.txt", "memset_probe.txt": "memset_probe.txt", "fill_probe.txt": "fill_probe.txt", "pmc_x1024.txt": "x1024_counters.txt", "build_probe.txt": "build_probe.txt", "pmc_build.txt": "build_counters.txt", "build_timeline_118.txt": "lbvh_build_timeline_69m.txt",
    }
    for s, d in names.items():
        p = os.path.join(src, s)
        if os.path.exists(p):
            text = [ln for ln in open(p, errors="replace") if "amdgpu.ids" not in ln]  # a harmless libdrm warning
            with open(os.path.join(dst, "%s_%s" % (a.prefix, d)), "w") as f:
                f.writelines(text)
        else:
            print("missing", p)
    tool = os.path.join(ROOT, "tools", "summarize_profiles.py")
    subprocess.check_call([sys.executable, tool, a.prefix, os.path.join(src, "stats"), "--fetch", os.path.join(src, "fetch"),
                           "--write", os.path.join(src, "write"), "--frames-per-launch", str(a.frames_per_launch)])
    subprocess.check_call([sys.executable, tool, a.prefix + "_ao64", os.path.join(src, "stats_ao")])
    if os.path.isdir(os.path.join(src, "stats_x1024")):
        subprocess.check_call([sys.executable, tool, a.prefix + "_x1024", os.path.join(src, "stats_x1024"), "--fetch",
                               os.path.join(src, "fetch_x1024"), "--write", os.path.join(src, "write_x1024"), "--workload",
                               "bunny_x1024_1080p", "--note", "one lane per triangle; the run also holds six scene builds"])
    if os.path.exists(os.path.join(src, "pmc_cache.txt")):
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "valu_summary.py"), os.path.join(src, "pmc_cache.txt"),
                               os.path.join(dst, a.prefix + "_valu.json")], stdout=subprocess.DEVNULL)
    shutil.rmtree(os.path.join(dst, "__pycache__"), ignore_errors=True)


if __name__ == "__main__":
    main()
