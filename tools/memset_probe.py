"""What this GPU's memory takes as plain fills, beside the renderer's own clear: hipMemsetAsync (through torch) and a
device-to-device copy over buffers far larger than the Infinity Cache.

    python tools/memset_probe.py [--gb 1.19]
"""
import argparse
import time

import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gb", type=float, default=1.19)
    a = ap.parse_args()
    n = int(a.gb * 1e9) // 4
    x = torch.empty(n, dtype=torch.int32, device="cuda")
    y = torch.empty(n, dtype=torch.int32, device="cuda")
    for name, fn, nbytes in (("fill (hipMemsetAsync, zero_)", lambda: x.zero_(), 4 * n),
                             ("fill (a value: fill_ kernel)", lambda: x.fill_(0x01020304), 4 * n),
                             ("copy (device to device)", lambda: y.copy_(x), 8 * n)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
        print("%-32s %.2f GB per call: %.3f ms, %.2f TB/s (bytes read + written)" % (name, 4 * n / 1e9, best, nbytes / best / 1e9))


if __name__ == "__main__":
    main()
