#!/usr/bin/env python3
"""Decode-kernel cost model: times h264_amd_k_decode_slices on batches of
1080p I-slices with different macroblock mixes (all I_16x16 without residual,
all I_NxN without residual, sparse / dense residual, I_PCM) and prints cycles
per macroblock per CU for each, so that the cost of a prediction-mode symbol,
an empty block and a coded coefficient can be told apart.

    python tools/bench_mix.py [--slices 4096] [--tier 1]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

MIXES = {
    "i16_empty": dict(pm_i16x16=1000, pm_cbp_zero=1000, density=0),
    "inxn_empty": dict(pm_i16x16=0, pm_cbp_zero=1000, density=0),
    "inxn_sparse": dict(pm_i16x16=0, pm_cbp_zero=0, density=1),
    "inxn_mid": dict(pm_i16x16=0, pm_cbp_zero=0, density=5),
    "inxn_dense": dict(pm_i16x16=0, pm_cbp_zero=0, density=12),
    "i16_mid": dict(pm_i16x16=1000, pm_cbp_zero=0, density=5),
    "pcm": dict(pm_pcm=1000),
    "bench": dict(pm_i16x16=280, pm_pcm=2, density=5, pm_escape=5,
                  pm_cbp_zero=50, pm_zero_pcm=100),
    # inter slices (generic instantiation of the kernel)
    "p_mid": dict(slice_type=5, idr=0, pm_skip=300, pm_intra_in_inter=100,
                  pm_i16x16=300, density=5, frame_num=1),
    "b_mid": dict(slice_type=6, idr=0, nal_ref_idc=0, pm_skip=300,
                  pm_intra_in_inter=100, pm_i16x16=300, density=5,
                  frame_num=1),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--slices", type=int, default=4096)
    ap.add_argument("--tier", type=int, default=1)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    import libh264_amd
    from libh264_amd import gen, Params
    from libh264_amd.ctypes_abi import GenSlice, TIER_NONE, TIER_T0, TIER_T1
    import orc
    threads = min(len(os.sched_getaffinity(0)), 16)
    eng = libh264_amd.Engine(0)
    cfg = gen.cfg_1080p()
    rec = orc.parse_nalus([gen.sps(cfg), gen.pps(cfg)], flags=0, want=0)
    params = Params()
    params.put_sps(rec.sps[0])
    epoch = params.put_pps(rec.pps[0])
    tier = {0: TIER_NONE, 1: TIER_T0, 2: TIER_T1}[args.tier]
    for name, kw in MIXES.items():
        if args.only and name not in args.only.split(","):
            continue
        slices = []
        for i in range(args.slices):
            s = GenSlice(slice_type=7, idr=1, nal_ref_idc=3, first_mb=0,
                         num_mbs=8160, seed=1 + i)
            for k, v in kw.items():
                setattr(s, k, v)
            slices.append(s)
        buf, offs, sizes = gen.batch(cfg, slices, threads=threads)
        d_buf = eng.upload(buf)
        descs = np.zeros(args.slices, dtype=[("offset", "<u8"),
                                             ("size", "<u4"), ("epoch", "<u4")])
        descs["offset"], descs["size"], descs["epoch"] = offs, sizes, epoch
        d_descs = eng.upload(descs)
        out = eng.decode_slices(params, d_buf, descs, reader_flags=1,
                                tier=tier, mb_capacity=8160, d_descs=d_descs)
        res = eng.fetch_results(out)
        assert all(res[i].status == 0 and res[i].mb_count == 8160
                   for i in range(0, args.slices, 97))
        ms = []
        for _ in range(args.reps):
            eng.decode_slices(params, d_buf, descs, reader_flags=1, tier=tier,
                              mb_capacity=8160, d_descs=d_descs, outputs=out)
            ms.append(eng.last_kernel_ms("decode"))
        k_ms = float(np.median(ms))
        mbs = 8160 * args.slices
        total = int(sizes.sum())
        print(json.dumps({
            "mix": name, "bits_per_mb": round(8.0 * total / mbs, 1),
            "kernel_ms": round(k_ms, 3),
            "mmb_per_s": round(mbs / k_ms / 1e3, 1),
            "gbit_per_s": round(8e-6 * total / k_ms, 2),
            # 256 CUs at ~2.4 GHz
            "cu_cycles_per_mb": round(k_ms * 1e-3 * 2.4e9 * 256 / mbs, 1),
        }), flush=True)
        del d_buf, d_descs, out
    eng.close()


if __name__ == "__main__":
    main()
