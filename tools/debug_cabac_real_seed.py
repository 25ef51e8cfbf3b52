#!/usr/bin/env python3
"""Development aid: one seed of tools/fuzz_cabac_real.py, the slices whose
results differ decoded again at the full-syntax tier; prints the first
macroblock record where kernel and oracle differ.
    python tools/debug_cabac_real_seed.py SEED"""
import os
import random
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import libh264_amd  # noqa: E402
import orc  # noqa: E402
import streams  # noqa: E402
from libh264_amd.ctypes_abi import TIER_T2, t2_stride  # noqa: E402
from test_cabac_real_streams import quirk  # noqa: E402
from test_gpu_decode import make_params  # noqa: E402
from test_gpu_fuzz import mutate  # noqa: E402
import fuzz_cabac_real as fr  # noqa: E402

seed = int(sys.argv[1])
clips = [fr.clip("cockatoo.mp4") + (3602,), fr.clip("realshort.mp4") + (302,)]
rnd = random.Random(seed)
head, nals, cap = clips[rnd.randrange(2)]
pick = [rnd.choice(nals) for _ in range(24)]
batch = [n if rnd.randrange(4) == 0 else mutate(n, rnd) for n in pick]
q = rnd.randrange(2) == 1
flags = 0x101 | (0x200 if q else 0)
print("quirk", q, "cap", cap)
eng = libh264_amd.Engine(0)
quirk(q)
ref = orc.parse_nalus(head + batch, flags=0x101)
quirk(False)
params = make_params(ref)
buf, offs, sizes = streams.pack_nals(batch, lead=seed % 4)
out = eng.decode_slices(params, eng.upload(buf),
                        [(o, z, params.epoch) for o, z in zip(offs, sizes)],
                        reader_flags=flags, tier=TIER_T2, mb_capacity=cap,
                        want_headers=True)
res = eng.fetch_results(out)
cat = ref.sps[0].chroma_format_idc
st = t2_stride(cat)
for i in range(len(batch)):
    r, g = ref.results[i], res[i]
    if (r.status, r.mb_count, r.checksum) == (g.status, g.mb_count, g.checksum):
        continue
    w = ref.mb_records(i)
    m = min(g.mb_count, r.mb_count)
    got = eng.fetch_mb(out, i, max(g.mb_count, 1))[:, :st]
    print("slice %d: status %d/%d mbs %d/%d type %d" % (
        i, g.status, r.status, g.mb_count, r.mb_count, ref.headers[i].slice_type))
    for mb in range(m):
        if not np.array_equal(got[mb], w[mb]):
            bad = np.argwhere(got[mb] != w[mb]).ravel()
            print(" first differing mb %d: bytes %s" % (mb, bad[:24].tolist()))
            print("  want head", bytes(w[mb, :64]).hex())
            print("  got  head", bytes(got[mb, :64]).hex())
            for b in bad[:12]:
                print("   byte %d: got %d want %d" % (b, got[mb, b], w[mb, b]))
            if mb:
                print("  previous mb head", bytes(w[mb - 1, :32]).hex())
            break
    else:
        print(" records equal for the first %d macroblocks" % m)
