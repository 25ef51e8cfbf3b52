#!/usr/bin/env python3
"""Bins of the C3 bench slices by kind, from the oracle's decoder (CPU): which
syntax elements the arithmetic decoder's time goes to.

    python tools/cabac_bin_mix.py [--slices 4]"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

# (first ctxIdx, last ctxIdx, name), Table 9-34
GROUPS = [
    (0, 10, "mb_type I / SI"), (11, 13, "mb_skip_flag P"), (14, 20, "mb_type P"),
    (21, 23, "sub_mb_type P"), (24, 26, "mb_skip_flag B"), (27, 35, "mb_type B"),
    (36, 39, "sub_mb_type B"), (40, 53, "mvd"), (54, 59, "ref_idx"),
    (60, 63, "mb_qp_delta"), (64, 67, "intra_chroma_pred_mode"),
    (68, 69, "intra pred mode"), (70, 72, "field flag"),
    (73, 84, "coded_block_pattern"), (85, 104, "coded_block_flag"),
    (105, 165, "significant_coeff_flag 4x4 (frame)"),
    (166, 226, "last_significant_coeff_flag 4x4 (frame)"),
    (227, 275, "coeff_abs_level_minus1 4x4"),
    (276, 276, "end_of_slice (ctx)"),
    (277, 398, "sig / last 4x4 (field)"),
    (399, 401, "transform_size_8x8_flag"),
    (402, 416, "significant_coeff_flag 8x8 (frame)"),
    (417, 425, "last_significant_coeff_flag 8x8 (frame)"),
    (426, 435, "coeff_abs_level_minus1 8x8"),
    (436, 459, "sig / last 8x8 (field)"),
    (460, 1023, "Cb / Cr of 4:4:4, coded_block_flag 8x8"),
    (1024, 1024, "bypass (signs, Exp-Golomb suffixes)"),
    (1025, 1025, "terminate"),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--slices", type=int, default=4)
    args = ap.parse_args()
    import bench_cabac
    import orc
    from libh264_amd import gen
    cfg, buf, offs, sizes, bins = bench_cabac.workload(args.slices, 4)
    nals = [bytes(buf[int(o):int(o) + int(z)]) for o, z in zip(offs, sizes)]
    hist = (C.c_uint64 * 1026).in_dll(orc.lib(), "orc_stat_cabac_hist")
    for i in range(1026):
        hist[i] = 0
    rec = orc.parse_nalus([gen.sps(cfg), gen.pps(cfg)] + nals, flags=0x101, want=0)
    assert all(r.status == 0 for r in rec.results)
    total = sum(hist)
    mbs = 8160.0 * args.slices
    out = {"slices": args.slices, "bins_per_mb": round(total / mbs, 1)}
    for lo, hi, name in GROUPS:
        v = sum(hist[lo:hi + 1])
        if v:
            out[name] = [round(v / mbs, 2), round(100.0 * v / total, 1)]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
