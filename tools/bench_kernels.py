#!/usr/bin/env python3
"""Side measurements for the kernels that are not bench.py's headline:
the Annex-B scan (BASELINE config 4: 64 MiB mixed byte stream) and the slice
write-back (config 5: re-emit parsed slices, byte-identical).  Prints one JSON
line per kernel with the achieved fraction of the HBM roofline.

    python tools/bench_kernels.py [--mib 64] [--slices 1024] [--reps 10]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0


def mixed_stream(target_bytes, threads, seed0=0):
    """GOPs of SPS/PPS/AUD + one IDR and P slices of 1080p, 3- and 4-byte
    start codes, a filler now and then."""
    from libh264_amd import gen
    from libh264_amd.ctypes_abi import GenSlice
    cfg = gen.cfg_1080p()
    n = max(8, target_bytes // 120000)
    slices = [GenSlice(slice_type=7 if k % 8 == 0 else 5,
                       idr=int(k % 8 == 0), nal_ref_idc=1, first_mb=0,
                       num_mbs=8160, frame_num=k % 16, density=5,
                       pm_i16x16=250, pm_skip=150, pm_intra_in_inter=200,
                       pm_escape=5, seed=seed0 + k + 1) for k in range(n)]
    buf, offs, sizes = gen.batch(cfg, slices, threads=threads)
    sps, pps, aud, fil = gen.sps(cfg), gen.pps(cfg), gen.aud(0), gen.filler(40)
    parts = []
    total = 0
    for k in range(n):
        if k % 8 == 0:
            parts += [b"\x00\x00\x00\x01", sps, b"\x00\x00\x00\x01", pps]
        parts += [b"\x00\x00\x00\x01", aud]
        parts.append(b"\x00\x00\x01" if k % 3 else b"\x00\x00\x00\x01")
        parts.append(buf[int(offs[k]):int(offs[k]) + int(sizes[k])].tobytes())
        if k % 5 == 0:
            parts += [b"\x00\x00\x01", fil, b"\x00"]
        total += int(sizes[k])
        if total >= target_bytes:
            break
    return b"".join(parts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mib", type=int, default=64)
    ap.add_argument("--slices", type=int, default=1024)
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()

    import libh264_amd
    from libh264_amd import gen, Params
    from libh264_amd.ctypes_abi import TIER_NONE
    import orc
    import streams
    threads = min(len(os.sched_getaffinity(0)), 16)
    eng = libh264_amd.Engine(0)

    # ---- scan ------------------------------------------------------------
    data = mixed_stream(args.mib << 20, threads)
    d_buf = eng.upload(np.frombuffer(data + b"\xff" * 64, dtype=np.uint8))
    cap = 1 << 16
    tab, count, consumed = eng.scan(d_buf, len(data), cap)
    ms, ms_all = [], []
    for _ in range(args.reps):
        eng.scan(d_buf, len(data), cap)
        ms.append(eng.last_kernel_ms("scan"))
        ms_all.append(eng.last_kernel_ms("scan_all"))
    k_ms = float(np.median(ms))
    # the oracle on a prefix (bounded) as checker of the table's head
    head = data[:4 << 20]
    ref = orc.parse_stream(head, flags=0, want=0)
    nref = len(ref.nals) - 1  # the last NAL of the prefix is cut short
    assert [(int(a), int(b)) for a, b in tab[:nref]] == \
        [(n.start, n.end) for n in ref.nals[:nref]]
    print(json.dumps({
        "kernel": "h264_amd_k_scan_mark", "workload": "C4: %.1f MiB mixed "
        "Annex-B stream, %d NAL units" % (len(data) / 2 ** 20, count),
        "kernel_ms": round(k_ms, 4),
        "mark_prefix_place_ms": round(float(np.median(ms_all)), 4),
        "algorithmic_bytes": len(data),
        "achieved_GBps": round(len(data) / k_ms / 1e6, 1),
        "frac_of_hbm_peak": round(len(data) / k_ms / 1e6 / HBM_PEAK_GBS, 4),
        "consumed": consumed}), flush=True)

    # ---- write-back ------------------------------------------------------
    sys.path.insert(0, ROOT)
    import bench
    cfg, buf, offs, sizes = bench.make_workload(args.slices, 17, threads)
    rec = orc.parse_nalus([gen.sps(cfg), gen.pps(cfg)], flags=0, want=0)
    params = Params()
    params.put_sps(rec.sps[0])
    epoch = params.put_pps(rec.pps[0])
    d_src = eng.upload(buf)
    descs = [(int(o), int(s), epoch) for o, s in zip(offs, sizes)]
    out = eng.decode_slices(params, d_src, descs, reader_flags=0,
                            tier=TIER_NONE, want_headers=True)
    capn = int(sizes.max()) + 64
    d_out, ooffs, osizes = eng.write_slices(params, d_src, out, capn)
    assert (osizes == sizes.astype(np.int32)).all()
    raw = eng.download(d_out)
    for i in range(0, args.slices, max(1, args.slices // 16)):
        a = raw[int(ooffs[i]):int(ooffs[i]) + int(sizes[i])]
        b = buf[int(offs[i]):int(offs[i]) + int(sizes[i])]
        assert np.array_equal(a, b), i
    ms = []
    for _ in range(args.reps):
        eng.write_slices(params, d_src, out, capn)
        ms.append(eng.last_kernel_ms("write"))
    k_ms = float(np.median(ms))
    total = int(sizes.sum())
    print(json.dumps({
        "kernel": "h264_amd_k_write_slices", "workload": "C5: %d parsed 1080p "
        "slices re-emitted, byte-identical" % args.slices,
        "kernel_ms": round(k_ms, 4),
        "algorithmic_bytes": 2 * total,
        "achieved_GBps": round(2 * total / k_ms / 1e6, 1),
        "frac_of_hbm_peak": round(2 * total / k_ms / 1e6 / HBM_PEAK_GBS, 4)}),
        flush=True)
    # ---- C3: CABAC P-slices ------------------------------------------------
    # reader side: header + raw extent only (the reference has no CABAC
    # decoder); writer side: h264_write_skipped_p_slice in CABAC, 8160 MBs
    from libh264_amd.ctypes_abi import Sps, Pps, SliceHeader
    sps, pps = Sps(), Pps()
    sps.profile_idc, sps.level_idc, sps.chroma_format_idc = 100, 40, 1
    sps.pic_order_cnt_type, sps.max_num_ref_frames = 2, 1
    sps.pic_width_in_mbs_minus1, sps.pic_height_in_map_units_minus1 = 119, 67
    sps.frame_mbs_only_flag = sps.direct_8x8_inference_flag = 1
    pps.entropy_coding_mode_flag = 1
    pps.deblocking_filter_control_present_flag = 1
    pps.transform_8x8_mode_flag = 1
    cparams = Params()
    cparams.put_sps(sps)
    cepoch = cparams.put_pps(pps)
    n = args.slices
    descs3, hdrs3 = [], []
    for k in range(n):
        sh = SliceHeader()
        sh.slice_type = 5
        sh.frame_num = k % 16
        sh.slice_qp_delta = (k % 21) - 10
        sh.cabac_init_idc = k % 3
        descs3.append((1, cepoch, 8160, 2, 1))
        hdrs3.append(sh)
    got = eng.encode_synth(cparams, descs3, hdrs3, 1 << 10)
    b = orc.Builder()
    for k in range(0, n, max(1, n // 8)):
        assert got[k] == b.nalu(2, 1, sps=sps, pps=pps, sh=hdrs3[k], kind=2,
                                mb_count=8160), k
    ms = []
    for _ in range(args.reps):
        eng.encode_synth(cparams, descs3, hdrs3, 1 << 10)
        ms.append(eng.last_kernel_ms("synth"))
    k_ms = float(np.median(ms))
    print(json.dumps({
        "kernel": "h264_amd_k_encode_synth", "workload": "C3 writer side: %d "
        "all-skip CABAC P-slices of 8160 MBs" % n,
        "kernel_ms": round(k_ms, 4),
        "mmb_per_s": round(8160 * n / k_ms / 1e3, 1),
        "bytes_out": int(sum(len(g) for g in got))}), flush=True)
    # reader side on what was just written
    buf3, offs3, sizes3 = streams.pack_nals(got)
    d3 = eng.upload(buf3)
    dd = [(int(o), int(s), cepoch) for o, s in zip(offs3, sizes3)]
    out3 = eng.decode_slices(cparams, d3, dd, reader_flags=1, tier=TIER_NONE,
                             want_headers=True)
    r3 = eng.fetch_results(out3)
    assert all(r3[k].status == 0 and (r3[k].flags & 4) and r3[k].mb_count == 0
               for k in range(n))
    ms = []
    for _ in range(args.reps):
        eng.decode_slices(cparams, d3, dd, reader_flags=1, tier=TIER_NONE,
                          want_headers=True, outputs=out3)
        ms.append(eng.last_kernel_ms("decode"))
    k_ms = float(np.median(ms))
    print(json.dumps({
        "kernel": "h264_amd_k_decode_slices", "workload": "C3 reader side: %d "
        "CABAC P-slices, header + raw extent (no CABAC decoder in the "
        "reference)" % n, "kernel_ms": round(k_ms, 4),
        "slices_per_s": round(n / k_ms * 1e3, 1)}), flush=True)
    # the engine's CABAC subset decoder on the same slices (an extension:
    # every mb_skip_flag / end_of_slice bin is arithmetic-decoded)
    from libh264_amd.ctypes_abi import TIER_T0
    out4 = eng.decode_slices(cparams, d3, dd, reader_flags=0x101, tier=TIER_T0,
                             mb_capacity=8160, want_headers=True)
    r4 = eng.fetch_results(out4)
    assert all(r4[k].status == 0 and r4[k].mb_count == 8160 for k in range(n))
    ref = orc.parse_nalus([b.nalu(3, 7, sps=sps), b.nalu(3, 8, sps=sps, pps=pps)]
                          + [got[k] for k in range(0, n, max(1, n // 8))],
                          flags=0x101, want=0)
    for j, k in enumerate(range(0, n, max(1, n // 8))):
        assert ref.results[j].checksum == r4[k].checksum, k
    ms = []
    for _ in range(args.reps):
        eng.decode_slices(cparams, d3, dd, reader_flags=0x101, tier=TIER_T0,
                          mb_capacity=8160, want_headers=True, outputs=out4)
        ms.append(eng.last_kernel_ms("cabac"))
    k_ms = float(np.median(ms))
    print(json.dumps({
        "kernel": "h264_amd_k_decode_cabac_subset", "workload": "C3 with the "
        "CABAC subset decoder: %d all-skip CABAC P-slices of 8160 MBs, every "
        "bin arithmetic-decoded, T0 records" % n, "kernel_ms": round(k_ms, 4),
        "mmb_per_s": round(8160 * n / k_ms / 1e3, 1),
        "bins_per_s": round(2 * 8160 * n / k_ms * 1e3, 1)}), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
