#!/usr/bin/env python3
"""End-to-end cost of the DROP-IN API (host buffer in, callbacks out) on a mixed
Annex-B stream: h264_reader_parse() = H2D copy + GPU scan + host pre-pass + one
batched slice decode + D2H of the records + callback replay on the caller's
thread.  The PCIe-inclusive number DESIGN.md quotes next to the HBM-resident
kernel rate of bench.py.

    python tools/bench_dropin.py [--mib 64] [--flags 1]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mib", type=int, default=64)
    ap.add_argument("--flags", type=int, default=1)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import bench_kernels
    import prod
    data = bench_kernels.mixed_stream(args.mib << 20,
                                      min(len(os.sched_getaffinity(0)), 16))
    L = prod.recorder()
    best = None
    for _ in range(args.reps):
        rec = L.rec_new(args.flags, 0)
        off = C.c_size_t()
        t0 = time.perf_counter()
        res = L.rec_parse_stream(rec, data, len(data), C.byref(off))
        dt = time.perf_counter() - t0
        n = C.c_size_t()
        L.rec_events(rec, C.byref(n))
        events = n.value
        L.rec_free(rec)
        assert res == 0 and off.value == len(data)
        best = dt if best is None else min(best, dt)
    print(json.dumps({
        "api": "h264_reader_parse (drop-in, host buffer, all callbacks)",
        "stream_mib": round(len(data) / 2 ** 20, 1), "reader_flags": args.flags,
        "callbacks": events, "seconds": round(best, 4),
        "mbit_per_s": round(8e-6 * len(data) / best, 1)}))


if __name__ == "__main__":
    main()
