#!/usr/bin/env python3
"""Development aid: decode generator CABAC slices on the GPU with T2 records and
print, per slice, the first record byte where kernel and oracle differ."""
import sys
import os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
import libh264_amd
import orc
import streams
from libh264_amd import gen
from libh264_amd.ctypes_abi import TIER_T2, t2_stride
from test_cabac_full import FLAGS, cabac_cfg, cabac_slice
from test_gpu_decode import make_params

cf = int(sys.argv[1]) if len(sys.argv) > 1 else 3
bd = int(sys.argv[2]) if len(sys.argv) > 2 else 0
kw = {}
for a in sys.argv[3:]:
    k, v = a.split("=")
    kw[k] = int(v)
eng = libh264_amd.Engine(0)
cfg = cabac_cfg(cf=cf, bd=bd)
slices = [cabac_slice(cfg, seed, (2, 0, 1)[seed % 3], **kw) for seed in range(18)]
head = [gen.sps(cfg), gen.pps(cfg)]
nals = [gen.slice_nal(cfg, s) for s in slices]
ref = orc.parse_nalus(head + nals, flags=FLAGS)
params = make_params(ref)
buf, offs, sizes = streams.pack_nals(nals)
d_buf = eng.upload(buf)
descs = [(o, s, params.epoch) for o, s in zip(offs, sizes)]
out = eng.decode_slices(params, d_buf, descs, reader_flags=FLAGS, tier=TIER_T2,
                        mb_capacity=64, want_headers=True)
res = eng.fetch_results(out)
st = t2_stride(cf)
for i in range(len(nals)):
    w = ref.mb_records(i)
    n = res[i].mb_count
    g = eng.fetch_mb(out, i, max(n, 1))[:, :st]
    line = "slice %2d type %d: status %d/%d mbs %d/%d" % (
        i, slices[i].slice_type, res[i].status, ref.results[i].status, n,
        ref.results[i].mb_count)
    m = min(n, len(w))
    if m and not np.array_equal(g[:m], w[:m]):
        bad = np.argwhere(g[:m] != w[:m])[0]
        mb = bad[0]
        line += " | first diff mb %d byte %d got %d want %d; mb_type %d cbp %#x flags %#x" % (
            mb, bad[1], g[mb, bad[1]], w[mb, bad[1]], w[mb, 0], w[mb, 1], w[mb, 2])
        if n > mb:
            line += " ; next mb want type %d" % (w[mb + 1, 0] if mb + 1 < len(w) else -1)
    elif n < len(w):
        line += " | records equal up to %d; next want: type %d cbp %#x flags %#x raw %d" % (
            n, w[n, 0], w[n, 1], w[n, 2], w[n, 3])
    print(line)
