#!/usr/bin/env python3
"""Development aid: the batch of tests/test_gpu_cabac.py::
test_damaged_and_foreign_cabac_data with T2 records; for every slice whose
result differs from the oracle's, status / macroblock counts and the first
differing record."""
import os
import random
import sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
import libh264_amd
import orc
import streams
import cabac_enc
from libh264_amd.ctypes_abi import TIER_T2, t2_stride
from test_cabac_subset import FLAGS, custom_i_slice, header, param_sets
from test_gpu_fuzz import mutate
from test_gpu_decode import make_params

rnd = random.Random(11)
sps, pps = param_sets(8, 6, 1)
b = orc.Builder()
head = [b.nalu(3, 7, sps=sps), b.nalu(3, 8, sps=sps, pps=pps)]
nals = []
for k in range(160):
    kind = k % 4
    if kind == 0:
        n = rnd.randrange(1, 49)
        mbs = [(rnd.randrange(4), rnd.randrange(4),
                rnd.choice([0, 0, 1, -2, 5])) for _ in range(n)]
        nal = custom_i_slice(b, sps, pps, mbs, 0, rnd.randrange(-5, 5), k)
    elif kind == 1:
        st = rnd.choice([5, 6])
        sh = header(st, rnd.randrange(0, 20), rnd.randrange(-5, 5),
                    rnd.randrange(3), k)
        nal = b.nalu(2, 1, sps=sps, pps=pps, sh=sh, kind=2,
                     mb_count=rnd.randrange(1, 28), cap=1 << 14)
    elif kind == 2:
        sh = header(7, 0, 0, 0, k)
        nal = b.nalu(3, 5, sps=sps, pps=pps, sh=sh, kind=1,
                     mb_count=rnd.randrange(1, 48), cap=1 << 14)
    else:
        st = rnd.choice([5, 6, 7])
        sh = header(st, 0, 0, rnd.randrange(3), k)
        hd = b.nalu(3 if st == 7 else 2, 5 if st == 7 else 1, sps=sps,
                    pps=pps, sh=sh, kind=1 if st == 7 else 2, mb_count=1,
                    cap=1 << 14)
        r = orc.parse_nalus(head + [hd], flags=0).results[0]
        nal = cabac_enc.escape_tail(
            hd[:r.rawdata_off],
            bytes(rnd.randrange(256) for _ in range(rnd.randrange(0, 60))))
    nals.append(mutate(nal, rnd) if (k % 8) >= 3 else nal)
eng = libh264_amd.Engine(0)
ref = orc.parse_nalus(head + nals, flags=FLAGS)
params = make_params(ref)
buf, offs, sizes = streams.pack_nals(nals)
d_buf = eng.upload(buf)
descs = [(o, s, params.epoch) for o, s in zip(offs, sizes)]
out = eng.decode_slices(params, d_buf, descs, reader_flags=FLAGS, tier=TIER_T2,
                        mb_capacity=200, want_headers=True)
res = eng.fetch_results(out)
st = t2_stride(1)
for i in range(len(nals)):
    rr = ref.results[i]
    if (res[i].status, res[i].mb_count, res[i].checksum) == \
            (rr.status, rr.mb_count, rr.checksum):
        continue
    w = ref.mb_records(i)
    n = res[i].mb_count
    g = eng.fetch_mb(out, i, max(n, 1))[:, :st]
    print("slice %d (%d bytes) status %d/%d mbs %d/%d" % (
        i, len(nals[i]), res[i].status, rr.status, n, rr.mb_count))
    m = min(n, len(w))
    for mb in range(max(m, 1)):
        if mb < len(w):
            print("  want mb %d: type %d cbp %#x flags %#x raw %d  head %s" % (
                mb, w[mb, 0], w[mb, 1], w[mb, 2], w[mb, 3],
                bytes(w[mb, :32]).hex()))
        if mb < n:
            print("  got  mb %d: type %d cbp %#x flags %#x raw %d  head %s" % (
                mb, g[mb, 0], g[mb, 1], g[mb, 2], g[mb, 3],
                bytes(g[mb, :32]).hex()))
    for mb in range(m, min(len(w), m + 3)):
        print("  want mb %d: type %d cbp %#x flags %#x raw %d" % (
            mb, w[mb, 0], w[mb, 1], w[mb, 2], w[mb, 3]))
