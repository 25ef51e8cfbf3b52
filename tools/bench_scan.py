#!/usr/bin/env python3
"""Annex-B scan kernel alone on a large synthetic byte stream: payload bytes
without zero pairs (what emulation prevention guarantees), NAL units of
`--nal` bytes on average with 3- and 4-byte start codes, plus the worst case
(byte soup rich in zeros).  The NAL table is checked against the positions the
stream was built from and, on a prefix, against the oracle.

    python tools/bench_scan.py [--mib 1024] [--nal 65536] [--reps 10]
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


def build(total, nal, seed):
    rng = np.random.default_rng(seed)
    data = rng.integers(1, 256, size=total, dtype=np.uint8)
    pos, expect = 0, []
    gaps = rng.integers(nal // 2, nal * 3 // 2 + 1, size=total // nal * 3 + 8)
    k = 0
    prev_start = None
    while pos + 8 < total:
        long_code = bool(gaps[k] & 1)
        if long_code:
            data[pos:pos + 4] = (0, 0, 0, 1)
        else:
            data[pos:pos + 3] = (0, 0, 1)
        if prev_start is not None:
            expect.append((prev_start, pos))
        prev_start = pos + (4 if long_code else 3)
        pos = prev_start + int(gaps[k])
        k += 1
    expect.append((prev_start, total))
    return data, expect


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mib", type=int, default=1024)
    ap.add_argument("--nal", type=int, default=65536)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--soup", type=int, default=1)
    args = ap.parse_args()

    import libh264_amd
    import orc
    eng = libh264_amd.Engine(0)
    total = args.mib << 20
    data, expect = build(total, args.nal, 7)
    buf = np.concatenate([data, np.full(64, 0xff, dtype=np.uint8)])
    d_buf = eng.upload(buf)
    cap = len(expect) + 16
    tab, count, consumed = eng.scan(d_buf, total, cap)
    assert count == len(expect), (count, len(expect))
    got = np.asarray(tab[:count], dtype=np.uint64).reshape(-1, 2)
    want = np.asarray(expect, dtype=np.uint64)
    assert np.array_equal(got, want), "NAL table differs from construction"
    ref = orc.parse_stream(data[:2 << 20].tobytes(), flags=0, want=0)
    nref = len(ref.nals) - 1
    assert [(int(a), int(b)) for a, b in got[:nref]] == \
        [(n.start, n.end) for n in ref.nals[:nref]]
    ms, ms_all = [], []
    for _ in range(args.reps):
        eng.scan(d_buf, total, cap)
        ms.append(eng.last_kernel_ms("scan"))
        ms_all.append(eng.last_kernel_ms("scan_all"))
    k_ms = float(np.median(ms))
    print(json.dumps({
        "kernel": "h264_amd_k_scan_mark",
        "workload": "%d MiB, payload without zero pairs, %d NAL units" %
                    (args.mib, count),
        "seg_override": os.environ.get("H264_AMD_SCAN_SEG"),
        "kernel_ms": round(k_ms, 4), "min_ms": round(min(ms), 4),
        "all_three_kernels_ms": round(float(np.median(ms_all)), 4),
        "algorithmic_bytes": total,
        "achieved_GBps": round(total / k_ms / 1e6, 1),
        "frac_of_hbm_peak": round(total / k_ms / 1e6 / HBM_PEAK_GBS, 4)}),
        flush=True)
    # h264_byte_stream_to_avcc on the same bytes (in place: a fresh copy per
    # repetition); checked against the construction: every four-byte code
    # carries the distance to the next one
    ms = []
    for _ in range(3):
        eng.upload(buf, d_buf)
        eng.byte_stream_to_avcc(d_buf, total)
        ms.append(eng.last_kernel_ms("avcc"))
    got = eng.download(d_buf, total)
    long_starts = np.array([s for s, _ in expect if s >= 4 and
                            not data[s - 4:s - 1].any() and data[s - 4 + 3] == 1],
                           dtype=np.int64)
    nxt = np.append(long_starts[1:] - 4, total)
    be = got[(long_starts[:, None] - 4 + np.arange(4)[None, :])].astype(np.uint64)
    lens = (be[:, 0] << 24) | (be[:, 1] << 16) | (be[:, 2] << 8) | be[:, 3]
    assert np.array_equal(lens, (nxt - long_starts).astype(np.uint64))
    k_ms = float(np.median(ms))
    print(json.dumps({
        "kernel": "h264_amd_byte_stream_to_avcc (scan mark/prefix/place in "
                  "four-byte-code mode + length writer)",
        "workload": "%d MiB, %d four-byte start codes" % (args.mib,
                                                          len(long_starts)),
        "ms": round(k_ms, 4), "algorithmic_bytes": total,
        "achieved_GBps": round(total / k_ms / 1e6, 1),
        "frac_of_hbm_peak": round(total / k_ms / 1e6 / HBM_PEAK_GBS, 4)}),
        flush=True)
    if args.soup:
        n = min(total, 16 << 20)
        rng = np.random.default_rng(3)
        soup = rng.choice(np.array([0, 0, 0, 1, 3, 7, 255], dtype=np.uint8),
                          size=n)
        d_soup = eng.upload(np.concatenate([soup, np.full(64, 0xff,
                                                          dtype=np.uint8)]))
        scap = n // 3 + 8
        tab, count, consumed = eng.scan(d_soup, n, scap)
        ref = orc.parse_stream(soup[:1 << 20].tobytes(), flags=0, want=0)
        nref = len(ref.nals) - 1
        assert [(int(a), int(b)) for a, b in tab[:nref]] == \
            [(x.start, x.end) for x in ref.nals[:nref]]
        ms = []
        for _ in range(args.reps):
            eng.scan(d_soup, n, scap)
            ms.append(eng.last_kernel_ms("scan"))
        k_ms = float(np.median(ms))
        print(json.dumps({
            "kernel": "h264_amd_k_scan_mark",
            "workload": "worst case: %d MiB byte soup, %d NAL units" %
                        (n >> 20, count),
            "kernel_ms": round(k_ms, 4),
            "achieved_GBps": round(n / k_ms / 1e6, 1)}), flush=True)


if __name__ == "__main__":
    main()
