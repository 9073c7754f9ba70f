"""VALU utilisation per kernel from the counter table tools/pmc_cache.sh writes (gpurun_out/pmc_cache.txt).

    python tools/valu_summary.py gpurun_out/TAG/pmc_cache.txt profiles/PREFIX_valu.json

Per workload and kernel, from one rocprofv3 --pmc pass (averages over the kernel's launches):
  issue_busy        SQ_ACTIVE_INST_VALU / SQ_BUSY_CU_CYCLES: share of the cycles in which a CU had waves during which
                    its SIMDs issued vector instructions (quad-cycles of the four SIMDs over CU cycles)
  lanes_active      SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU): share of the 64 lanes active per instruction
  frac_of_peak      4 * SQ_ACTIVE_INST_VALU * lanes_active / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs): vector
                    lane-cycles used over what the chip offers while the kernel runs (GRBM_GUI_ACTIVE sums the 8 XCDs)
  achieved_tflops   frac_of_peak * 157.3 (every lane-cycle counted as a fused multiply-add: an upper bound on the
                    arithmetic actually done)
"""
import json
import re
import sys

PEAK = 157.3


def main():
    src, dst = sys.argv[1], sys.argv[2]
    rows = {}
    for line in open(src):
        m = re.match(r"(\S+)/(\S+)/(L2|SQ)\s+(.*)", line)
        if not m:
            continue
        wl, kernel, kind, rest = m.groups()
        vals = {k: float(v) for k, v in re.findall(r"(\w+)=([0-9.eE+-]+)", rest)}
        rows.setdefault(wl, {}).setdefault(kernel, {}).update(vals)
    out = {}
    for wl, kernels in rows.items():
        for kernel, v in kernels.items():
            if "SQ_ACTIVE_INST_VALU" not in v or not v.get("SQ_BUSY_CU_CYCLES"):
                continue
            act, busy, thr, gui = v["SQ_ACTIVE_INST_VALU"], v["SQ_BUSY_CU_CYCLES"], v["SQ_THREAD_CYCLES_VALU"], v["GRBM_GUI_ACTIVE"]
            lanes = thr / (64.0 * act)
            frac = 4.0 * act * lanes / (gui / 8.0 * 256 * 4)
            rec = {"issue_busy": act / busy, "lanes_active": lanes, "frac_of_peak": frac, "achieved_tflops": frac * PEAK,
                   "peak_tflops": PEAK, "valu_instructions_per_launch": v.get("SQ_INSTS_VALU"),
                   "method": "rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"}
            if v.get("TCC_REQ_sum"):
                rec["l2_hit_rate"] = v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"])
            out.setdefault(wl, {})[kernel] = rec
    json.dump({"workloads": out}, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
