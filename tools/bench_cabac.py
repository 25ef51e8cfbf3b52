#!/usr/bin/env python3
"""The CABAC slice-data kernel (h264_amd_k_decode_cabac) on BASELINE config 3 with
real content: n 1080p High-profile CABAC P-slices from the generator's own
arithmetic encoder, one launch per rep, kernel time from HIP events.

    python tools/bench_cabac.py [--slices 4096] [--reps 5] [--st 5] [--density 5]

Prints one JSON line: ms, bins, bins/s, macroblocks/s, Mbit/s.  Every slice must
come back with status 0 and 8160 macroblocks; the digest is a hash over all
result records (compare builds with it).  H264_AMD_ENGINE_LIB selects another
build of the engine (tools/build_variant.sh)."""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def workload(n, cores, st=5, density=5, skip=250, seed0=1000):
    from libh264_amd import gen
    from libh264_amd.ctypes_abi import GenSlice
    cfg = gen.cfg_1080p(profile_idc=100, entropy_coding_mode_flag=1,
                        transform_8x8_mode_flag=1, max_num_ref_frames=2,
                        num_ref_idx_l0_default_minus1=1)
    slices = [GenSlice(slice_type=st, idr=int(st % 5 == 2), nal_ref_idc=2,
                       num_mbs=8160, density=density, frame_num=k % 16,
                       pm_i16x16=300, pm_skip=skip, pm_intra_in_inter=60,
                       pm_t8x8=300, pm_escape=5, pm_cbp_zero=150,
                       cabac_syntax=1, cabac_init_idc=k % 3,
                       slice_qp_delta=(k % 21) - 10, seed=seed0 + k)
              for k in range(n)]
    gen.cabac_bins(reset=True)
    bufs, offs, sizes, pos = [], [], [], 0
    for k in range(0, n, 512):
        b, o, z = gen.batch(cfg, slices[k:k + 512], threads=cores)
        pad = (-len(b)) % 16
        bufs.append(np.concatenate([b, np.zeros(pad, dtype=np.uint8)]))
        offs.append(o + np.uint64(pos))
        sizes.append(z)
        pos += len(b) + pad
    buf = np.concatenate(bufs + [np.full(64, 0xff, np.uint8)])
    return cfg, buf, np.concatenate(offs), np.concatenate(sizes), gen.cabac_bins()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--slices", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--st", type=int, default=5)
    ap.add_argument("--density", type=int, default=5)
    ap.add_argument("--skip", type=int, default=250)
    args = ap.parse_args()
    import libh264_amd
    import orc
    from libh264_amd import gen, Params
    from libh264_amd.ctypes_abi import TIER_T0
    cores = min(len(os.sched_getaffinity(0)), 16)
    n = args.slices
    cfg, buf, offs, sizes, bins = workload(n, cores, args.st, args.density,
                                           args.skip)
    head = orc.parse_nalus([gen.sps(cfg), gen.pps(cfg)], flags=0, want=0)
    params = Params()
    params.put_sps(head.sps[0])
    epoch = params.put_pps(head.pps[0])
    order = np.arange(n)  # the engine orders the launch itself
    descs = np.zeros(n, dtype=[("offset", "<u8"), ("size", "<u4"),
                               ("epoch", "<u4")])
    descs["offset"], descs["size"], descs["epoch"] = offs[order], sizes[order], epoch
    eng = libh264_amd.Engine(0)
    d_buf, d_descs = eng.upload(buf), eng.upload(descs)
    out = eng.decode_slices(params, d_buf, descs, reader_flags=0x101,
                            tier=TIER_T0, mb_capacity=8160, want_headers=True,
                            d_descs=d_descs)
    ms = []
    for _ in range(args.reps):
        eng.decode_slices(params, d_buf, descs, reader_flags=0x101,
                          tier=TIER_T0, mb_capacity=8160, want_headers=True,
                          d_descs=d_descs, outputs=out)
        ms.append(eng.last_kernel_ms("cabac"))
    res = eng.fetch_results(out)
    bad = [k for k in range(n) if res[k].status != 0 or res[k].mb_count != 8160]
    h = hashlib.sha256()
    for k in range(n):
        h.update(bytes(res[k]))
    k_ms = float(np.median(ms))
    total = int(sizes.sum())
    print(json.dumps({
        "slices": n, "slice_type": args.st, "bad": len(bad),
        "bits_per_mb": round(8.0 * total / (8160.0 * n), 1),
        "bins_per_mb": round(bins / (8160.0 * n), 1),
        "ms": round(k_ms, 3), "ms_all": [round(m, 2) for m in ms],
        "bins_per_s": round(1e3 * bins / k_ms, 1),
        "macroblocks_per_s": round(8160e3 * n / k_ms, 1),
        "mbit_per_s": round(8e-3 * total / k_ms, 1),
        "digest": h.hexdigest()[:16],
        "lib": os.path.basename(os.environ.get("H264_AMD_ENGINE_LIB", "")) or
        "in-tree"}))


if __name__ == "__main__":
    main()
