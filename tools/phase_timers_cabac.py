#!/usr/bin/env python3
"""Where the CABAC kernel's cycles go, phase by phase (GPU box):

    tools/build_variant.sh ctimers -DCS_PHASE_TIMERS -Wno-inline-asm \
        -mllvm -structurizecfg-skip-uniform-regions
    python tools/phase_timers_cabac.py [--slices 4096]

The diagnostic build sums s_memtime differences between marks per phase and
leaves them in the first 64 T0 words of every slice (cabac_slice.hip,
CS_PHASE_TIMERS).  Printed per macroblock: cycles and calls per phase, the cost
of a mark (phase 20) taken off.  With 4096 slices the waves share their SIMD
four ways: the cycles are wall cycles of a wave, i.e. about four times what the
phase occupies the SIMD.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

PHASES = ["macroblock top: priority, neighbours", "mb_skip_flag",
          "skipped macroblock: record, reset", "mb_type",
          "derived variables, mvd / ref_idx grids opened",
          "sub_mb_type (inter) / prediction modes, chroma mode (intra)",
          "ref_idx", "mvd", "(intra: modes; inter: nothing)",
          "coded_block_pattern, transform flag, record cleared",
          "mb_qp_delta, coded_block_flag grids opened",
          "block loop, coded_block_flag", "significance map",
          "levels, signs, coefficient register", "block store, nz, grid",
          "behind the blocks", "record head", "record emitted",
          "row context", "end_of_slice", "(mark cost)"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--slices", type=int, default=4096)
    ap.add_argument("--lib", default=os.path.join(
        ROOT, "libh264_amd/lib/variants/libengine_ctimers.so"))
    args = ap.parse_args()
    os.environ["H264_AMD_ENGINE_LIB"] = os.path.abspath(args.lib)
    import bench_cabac
    import libh264_amd
    import orc
    from libh264_amd import gen, Params
    from libh264_amd.ctypes_abi import TIER_T0
    n = args.slices
    cfg, buf, offs, sizes, bins = bench_cabac.workload(n, 16)
    head = orc.parse_nalus([gen.sps(cfg), gen.pps(cfg)], flags=0, want=0)
    params = Params()
    params.put_sps(head.sps[0])
    epoch = params.put_pps(head.pps[0])
    descs = np.zeros(n, dtype=[("offset", "<u8"), ("size", "<u4"),
                               ("epoch", "<u4")])
    descs["offset"], descs["size"], descs["epoch"] = offs, sizes, epoch
    eng = libh264_amd.Engine(0)
    d_buf, d_descs = eng.upload(buf), eng.upload(descs)
    out = None
    for _ in range(2):
        out = eng.decode_slices(params, d_buf, descs, reader_flags=0x101,
                                tier=TIER_T0, mb_capacity=8160,
                                want_headers=True, d_descs=d_descs, outputs=out)
    ms = eng.last_kernel_ms("cabac")
    NP = len(PHASES)
    acc = np.zeros(32)
    cnt = np.zeros(32)
    take = min(n, 64)
    for i in range(0, n, max(1, n // take)):
        w = eng.fetch_mb(out, i, 64).reshape(-1).view(np.uint32)[:64]
        acc += w[:32]
        cnt += w[32:64]
    take = len(range(0, n, max(1, n // take)))
    mbs = 8160.0 * take
    mark = acc[NP - 1] / max(cnt[NP - 1], 1)
    print("kernel %.2f ms, %d slices, %.1f bins per macroblock; a mark costs "
          "%.1f cycles" % (ms, n, bins / (8160.0 * n), mark))
    total = 0.0
    rows = []
    for k in range(NP - 1):
        net = acc[k] - mark * cnt[k]
        total += net
        rows.append((k, PHASES[k], net / mbs, cnt[k] / mbs, net / max(cnt[k], 1)))
    for k, name, c, calls, per in rows:
        print("%2d %-58s %8.1f cyc/MB %5.1f%%  %6.2f calls/MB %7.1f cyc/call"
              % (k, name, c, 100.0 * c * mbs / total, calls, per))
    print("sum %.1f cycles per macroblock (marks taken off)" % (total / mbs))
    print(json.dumps({"slices": n, "kernel_ms": ms, "mark_cycles": mark,
                      "phases": {PHASES[k]: round(c, 1) for k, _, c, _, _ in rows}}))


if __name__ == "__main__":
    main()
