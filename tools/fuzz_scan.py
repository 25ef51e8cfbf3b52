#!/usr/bin/env python3
"""Byte-level fuzz of the Annex-B scan kernels (both modes) against the oracle:
random lengths, zero densities from "entropy coded" to "byte soup", patterns
planted on step / block / segment edges, several forced segment sizes.

    python tools/fuzz_scan.py [first_seed] [count]
"""
import ctypes as C
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import libh264_amd  # noqa: E402
import orc  # noqa: E402

PATTERNS = [b"\x00\x00\x01", b"\x00\x00\x00\x01", b"\x00\x00\x00",
            b"\x00\x00\x03", b"\x00\x00\x00\x00\x01", b"\x00\x00\x02",
            b"\x00\x01", b"\x00\x00\x00\x00\x00\x00\x01", b"\x00"]


def make(rnd):
    n = rnd.choice([rnd.randrange(1, 64), rnd.randrange(64, 5000),
                    rnd.randrange(5000, 70000), rnd.randrange(70000, 600000)])
    mode = rnd.randrange(4)
    if mode == 0:      # byte soup
        alpha = [0, 0, 0, 1, 3, 2, 255, 0x65]
        d = bytearray(rnd.choice(alpha) for _ in range(min(n, 60000)))
    else:               # payload with zero density 0, 1/256 or 1/16
        p0 = (0, 1, 16)[mode - 1]
        a = np.random.default_rng(rnd.randrange(1 << 30)).integers(
            1, 256, size=n, dtype=np.uint8)
        if p0:
            z = np.random.default_rng(rnd.randrange(1 << 30)).integers(
                0, 256, size=n) < p0
            a[z] = 0
        d = bytearray(a.tobytes())
        for _ in range(rnd.randrange(0, 40)):
            pat = rnd.choice(PATTERNS)
            if len(d) > len(pat):
                p = rnd.randrange(0, len(d) - len(pat))
                d[p:p + len(pat)] = pat
    for edge in (1024, 4096, 8192, 16384, 32768, 65536):
        if len(d) > edge + 8 and rnd.randrange(3) == 0:
            pat = rnd.choice(PATTERNS)
            p = edge - rnd.randrange(0, len(pat) + 2)
            d[p:p + len(pat)] = pat
    if rnd.randrange(4) == 0 and len(d) > 8:
        d[-rnd.randrange(1, 6):] = rnd.choice(PATTERNS)[:5]
    return bytes(d)


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    O = orc.lib()
    O.orc_byte_stream_to_avcc.argtypes = [C.c_void_p, C.c_size_t]
    engines = []
    for seg in (None, 4096, 8192, 65536):
        if seg is None:
            os.environ.pop("H264_AMD_SCAN_SEG", None)
        else:
            os.environ["H264_AMD_SCAN_SEG"] = str(seg)
        engines.append((seg, libh264_amd.Engine(0)))
    bad = 0
    for seed in range(first, first + count):
        rnd = random.Random(seed)
        data = make(rnd)
        ref = orc.parse_stream(data, flags=0, want=0)
        want = [(n.start, n.end) for n in ref.nals]
        avcc = C.create_string_buffer(data, len(data))
        O.orc_byte_stream_to_avcc(avcc, len(data))
        seg, eng = engines[seed % len(engines)]
        buf = np.frombuffer(data + b"\xee" * 64, dtype=np.uint8)
        d_buf = eng.upload(buf)
        tab, cnt, consumed = eng.scan(d_buf, len(data), len(data) // 3 + 8)
        got = [(int(a), int(b)) for a, b in tab]
        msg = None
        if cnt != len(want) or got != want or consumed != ref.consumed:
            msg = "scan differs (%d vs %d units, consumed %d vs %d)" % (
                cnt, len(want), consumed, ref.consumed)
        else:
            eng.byte_stream_to_avcc(d_buf, len(data))
            out = eng.download(d_buf, len(data) + 64).tobytes()
            if out[:len(data)] != avcc.raw or out[len(data):] != b"\xee" * 64:
                msg = "avcc differs"
        d_buf.free()
        if msg:
            bad += 1
            print("seed %d len %d seg %s: %s" % (seed, len(data), seg, msg),
                  flush=True)
        if (seed - first) % 100 == 99:
            print("... %d seeds, %d divergences" % (seed - first + 1, bad),
                  flush=True)
    print("seeds %d..%d divergences: %d" % (first, first + count - 1, bad))


if __name__ == "__main__":
    main()
