#!/usr/bin/env python3
"""The CABAC slice-data kernel on REAL content: the slices of an MP4 file (by
default imageio's cockatoo.mp4 where the image has it: x264 High 4:4:4
Predictive, 1280 x 720, I / P / B), the whole clip repeated until the launch
fills the chip.

    python tools/bench_cabac_clip.py [file.mp4] [--copies 15] [--reps 5]

Prints one JSON line: slices, bits and bins per macroblock (bins from the
oracle's decoder), kernel ms, bins/s, macroblocks/s, Mbit/s.  Every slice must
come back complete."""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def default_clip():
    import glob
    return (glob.glob("/opt/conda/lib/python3*/site-packages/imageio/"
                      "resources/images/cockatoo.mp4") + [""])[0]


def measure(path, copies=15, reps=5, eng=None):
    """the dictionary main() prints; None if there is no such file"""
    import libh264_amd
    import mp4_demux
    import orc
    import streams
    from libh264_amd import Params
    from libh264_amd.ctypes_abi import TIER_T0
    if not path or not os.path.isfile(path):
        return None

    class A:
        pass
    args = A()
    args.copies, args.reps = copies, reps
    sps, pps, samples = mp4_demux.video_nals(path)
    head = [bytes(x) for x in sps] + [bytes(x) for x in pps]
    nals = [bytes(n) for s in samples for n in s if (n[0] & 31) in (1, 5)]
    # x264 before build 151 and 4:4:4: the stream needs the old rule
    pre151 = False
    for s in samples[:1]:
        for n in s:
            i = bytes(n).find(b"x264 - core ")
            if (n[0] & 31) == 6 and i >= 0:
                pre151 = int(bytes(n)[i + 12:i + 16].split()[0]) < 151
    quirk = C.c_int.in_dll(orc.lib(), "orc_quirk_x264_444_cbf")
    bins = C.c_uint64.in_dll(orc.lib(), "orc_stat_cabac_bins")
    quirk.value = int(pre151)
    bins.value = 0
    ref = orc.parse_nalus(head + nals, flags=0x101, want=0)
    quirk.value = 0
    nb = bins.value
    spsr = ref.sps[0]
    pic = (spsr.pic_width_in_mbs_minus1 + 1) * \
        (spsr.pic_height_in_map_units_minus1 + 1)
    assert all((r.status, r.mb_count) == (0, pic) for r in ref.results)
    params = Params()
    params.put_sps(ref.sps[0])
    epoch = params.put_pps(ref.pps[0])
    buf, offs, sizes = streams.pack_nals(nals * args.copies)
    n = len(offs)
    descs = np.zeros(n, dtype=[("offset", "<u8"), ("size", "<u4"),
                               ("epoch", "<u4")])
    descs["offset"], descs["size"], descs["epoch"] = offs, sizes, epoch
    if eng is None:
        eng = libh264_amd.Engine(0)
    d_buf, d_descs = eng.upload(buf), eng.upload(descs)
    flags = 0x101 | (0x200 if pre151 else 0)
    out = eng.decode_slices(params, d_buf, descs, reader_flags=flags,
                            tier=TIER_T0, mb_capacity=pic, want_headers=True,
                            d_descs=d_descs)
    ms = []
    for _ in range(args.reps):
        eng.decode_slices(params, d_buf, descs, reader_flags=flags,
                          tier=TIER_T0, mb_capacity=pic, want_headers=True,
                          d_descs=d_descs, outputs=out)
        ms.append(eng.last_kernel_ms("cabac"))
    res = eng.fetch_results(out)
    bad = sum(1 for k in range(n)
              if (res[k].status, res[k].mb_count, res[k].checksum) !=
              (0, pic, ref.results[k % len(nals)].checksum))
    k_ms = float(np.median(ms))
    total = int(np.asarray(sizes, dtype=np.int64).sum())
    types = [h.slice_type % 5 for h in ref.headers]
    return {
        "clip": os.path.basename(path), "x264_before_151": pre151,
        "picture_mbs": pic, "slices": n, "copies": args.copies,
        "slice_types": {"P": types.count(0), "B": types.count(1),
                        "I": types.count(2)},
        "bad": bad,
        "bits_per_mb": round(8.0 * total / (float(pic) * n), 2),
        "bins_per_mb": round(nb / (float(pic) * len(nals)), 2),
        "ms": round(k_ms, 3),
        "bins_per_s": round(1e3 * nb * args.copies / k_ms, 1),
        "macroblocks_per_s": round(1e3 * pic * n / k_ms, 1),
        "mbit_per_s": round(8e-3 * total / k_ms, 1),
        "pictures_per_s": round(1e3 * n / k_ms, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("file", nargs="?", default="")
    ap.add_argument("--copies", type=int, default=15)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    out = measure(args.file or default_clip(), args.copies, args.reps)
    if out is None:
        raise SystemExit("no clip (give an MP4 file with an AVC track)")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
