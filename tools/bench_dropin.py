#!/usr/bin/env python3
"""End-to-end cost of the DROP-IN API (host buffer in, callbacks out) on a mixed
Annex-B stream: h264_reader_parse() = H2D copy + GPU scan + host pre-pass + one
batched slice decode + D2H of the records + callback replay on the caller's
thread.  The PCIe-inclusive number DESIGN.md quotes next to the HBM-resident
kernel rate of bench.py.

    python tools/bench_dropin.py [--mib 64] [--flags 1]
    python tools/bench_dropin.py --per-nal

--per-nal: latency of ONE h264_reader_parse_nalu() call on a 1080p slice NAL
unit (the per-frame pattern of SURVEY.md 8(b)): flags = 0 on the host header
path, flags = 0 forced through the kernel (H264_AMD_NO_HOST_HEADERS), and
SLICE_DATA (upload + one-wave launch + download + callbacks).
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


def per_nal(prod):
    import numpy as np
    from libh264_amd import gen
    cfg = gen.cfg_1080p()
    slices = [gen.GenSlice(slice_type=7 if k % 8 == 0 else 5,
                           idr=int(k % 8 == 0), nal_ref_idc=1, first_mb=0,
                           num_mbs=8160, frame_num=k % 16, density=5,
                           pm_i16x16=250, pm_skip=150, pm_intra_in_inter=200,
                           seed=k + 1) for k in range(16)]
    nals = [gen.slice_nal(cfg, s) for s in slices]
    L = prod.recorder()

    def run(flags, reps, env=None):
        if env:
            os.environ[env] = "1"
        rec = L.rec_new(flags, 0)
        for ps in (gen.sps(cfg), gen.pps(cfg)):
            b = C.create_string_buffer(bytes(ps), len(ps))
            assert L.rec_parse_nalu(rec, C.addressof(b), C.addressof(b),
                                    len(ps)) == 0
        bufs = [C.create_string_buffer(bytes(n), len(n)) for n in nals]
        times = []
        for r in range(reps):
            b = bufs[r % len(bufs)]
            t0 = time.perf_counter()
            res = L.rec_parse_nalu(rec, C.addressof(b), C.addressof(b),
                                   len(nals[r % len(bufs)]))
            times.append(time.perf_counter() - t0)
            assert res == 0
        L.rec_free(rec)
        if env:
            del os.environ[env]
        t = np.array(times[len(times) // 5:])  # first calls warm things up
        return {"median_us": round(1e6 * float(np.median(t)), 1),
                "p90_us": round(1e6 * float(np.percentile(t, 90)), 1),
                "calls": len(t)}

    out = {
        "api": "h264_reader_parse_nalu, one 1080p slice NAL unit per call "
               "(mean %.0f KB), callbacks recorded" %
               (sum(len(n) for n in nals) / len(nals) / 1e3),
        "flags0_host_header_path": run(0, 2000),
        "flags0_through_the_kernel": run(0, 300, "H264_AMD_NO_HOST_HEADERS"),
        "slice_data_8160_mb_callbacks": run(1, 40),
    }
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mib", type=int, default=64)
    ap.add_argument("--flags", type=int, default=1)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--per-nal", action="store_true")
    args = ap.parse_args()
    import bench_kernels
    import prod
    if args.per_nal:
        return per_nal(prod)
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
