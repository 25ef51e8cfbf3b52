#!/usr/bin/env python3
"""Where the decode kernel's cycles go, phase by phase (GPU box):

    tools/build_variant.sh timers -DSD_PHASE_TIMERS -mllvm -structurizecfg-skip-uniform-regions
    python tools/phase_timers.py [--slices 512]

The diagnostic build sums s_memtime differences between marks per phase and
leaves them in the first 32 T0 words of every slice (slice_decode.hip,
SD_PHASE_TIMERS).  Printed per macroblock: cycles and calls per phase, the
cost of a mark (phase 15) taken off.  One wave per SIMD at --slices <= 1024:
latencies, not throughput.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

PHASES = ["loop top: sync, overrun", "neighbours, clear, mb_type",
          "mb_pred (intra modes, chroma mode)", "coded_block_pattern",
          "coeff_token", "trailing ones + levels", "total_zeros",
          "run_before", "values + store", "mb_qp_delta + block set-up",
          "empty-run pass (taken)", "block pick: sync, norm, run test, nC",
          "nz bookkeeping", "end of blocks: store, overrun",
          "summary, emit, next_mb", "(mark cost)"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--slices", type=int, default=512)
    ap.add_argument("--lib", default=os.path.join(
        ROOT, "libh264_amd/lib/variants/libengine_timers.so"))
    args = ap.parse_args()
    os.environ["H264_AMD_ENGINE_LIB"] = os.path.abspath(args.lib)
    import bench
    import libh264_amd
    from libh264_amd import gen, Params, shard
    from libh264_amd.ctypes_abi import TIER_T0
    import orc
    n = args.slices
    eng = libh264_amd.Engine(0)
    cfg, buf, offs, sizes = bench.make_workload(n, 1, 16)
    rec = orc.parse_nalus([gen.sps(cfg), gen.pps(cfg)], flags=0, want=0)
    params = Params()
    params.put_sps(rec.sps[0])
    epoch = params.put_pps(rec.pps[0])
    descs = np.zeros(n, dtype=[("offset", "<u8"), ("size", "<u4"),
                               ("epoch", "<u4")])
    descs["offset"], descs["size"], descs["epoch"] = offs, sizes, epoch
    d_buf, d_descs = eng.upload(buf), eng.upload(descs)
    out = None
    for _ in range(2):
        out = eng.decode_slices(params, d_buf, descs, reader_flags=1,
                                tier=TIER_T0, mb_capacity=8160,
                                d_descs=d_descs, outputs=out)
    ms = eng.last_kernel_ms("decode")
    acc = np.zeros(16)
    cnt = np.zeros(16)
    take = min(n, 64)
    mhz = []
    wall = []
    for i in range(take):
        w = eng.fetch_mb(out, i, 34).reshape(-1).view(np.uint32)[:34]
        if w[33]:
            mhz.append(100.0 * float(w[32]) / float(w[33]))
            wall.append(float(w[33]) / 1e5)
        acc += w[:16]
        cnt += w[16:32]
    # when did the waves start?  (word 34: 100 MHz tick at the start of the
    # slice data; a sample over the whole batch)
    starts, walls = [], []
    for i in range(0, n, max(1, n // 512)):
        w = eng.fetch_mb(out, i, 35).reshape(-1).view(np.uint32)[:35]
        starts.append(int(w[34]))
        walls.append((float(w[33]) / 1e5, i))
    ws = sorted(walls)
    print("own wall time over the slice data, sample of %d waves: min %.2f "
          "median %.2f 90 %% %.2f max %.2f ms (slice %d)" %
          (len(ws), ws[0][0], ws[len(ws) // 2][0], ws[int(len(ws) * 0.9)][0],
           ws[-1][0], ws[-1][1]))
    by_q = [np.mean([w for w, i in walls if i * 4 // n == q]) for q in range(4)]
    print("mean by quarter of the batch: " + " ".join("%.2f" % x for x in by_q))
    s0 = min(starts)
    late = sorted((x - s0) / 1e5 for x in starts)
    print("start of the slice data after the first wave's, ms: median %.2f, "
          "90 %% %.2f, max %.2f" % (late[len(late) // 2],
                                  late[int(len(late) * 0.9)], late[-1]))
    mbs = 8160.0 * take
    mark = acc[15] / max(cnt[15], 1)
    print("kernel %.2f ms, %d slices; a mark costs %.1f cycles" %
          (ms, n, mark))
    if mhz:
        print("shader clock over the slice data (s_memtime ticks per 100 MHz "
              "tick of s_memrealtime): %.0f MHz (min %.0f, max %.0f)" %
              (np.mean(mhz), min(mhz), max(mhz)))
        print("a wave's own wall time over its slice data: mean %.2f ms, "
              "max %.2f ms" % (np.mean(wall), max(wall)))
    total = 0.0
    rows = []
    for k in range(15):
        net = acc[k] - mark * cnt[k]
        total += net
        rows.append((k, PHASES[k], net / mbs, cnt[k] / mbs,
                     net / max(cnt[k], 1)))
    for k, name, c, calls, per in rows:
        print("%2d %-40s %8.1f cyc/MB %5.1f%%  %6.2f calls/MB %7.1f cyc/call"
              % (k, name, c, 100.0 * c * mbs / total, calls, per))
    print("sum %.1f cycles per macroblock (marks taken off)" % (total / mbs))
    print(json.dumps({"slices": n, "kernel_ms": ms, "mark_cycles": mark,
                      "phases": {PHASES[k]: round(c, 1)
                                 for k, _, c, _, _ in rows}}))


if __name__ == "__main__":
    main()
