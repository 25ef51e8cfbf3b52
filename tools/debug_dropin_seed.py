#!/usr/bin/env python3
"""Development aid: one seed of tools/fuzz_dropin.py's hostile-parameter mode;
prints the first callback events where the drop-in library and the oracle
differ.   python tools/debug_dropin_seed.py SEED"""
import os
import random
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import fuzz_dropin as fd  # noqa: E402

seed = int(sys.argv[1])
# the tool's own generator, with the comparison opened up
import fuzz_params as fp  # noqa: E402
import orc  # noqa: E402
import streams  # noqa: E402
from libh264_amd import gen  # noqa: E402
from test_gpu_fuzz import mutate  # noqa: E402
prod = fd.prod
rnd = random.Random(seed)
cf, bd = rnd.choice([(1, 0), (1, 0), (0, 0), (2, 0), (3, 0), (1, 2)])
cfg = streams.small_cfg(cf=cf, bd=bd, transform_8x8_mode_flag=rnd.randrange(2),
                        weighted_pred_flag=rnd.randrange(2),
                        weighted_bipred_idc=rnd.randrange(3))
base = orc.parse_nalus([gen.sps(cfg), gen.pps(cfg)])
b = orc.Builder()
total = cfg.width_mbs * cfg.height_mbs
nals = [gen.sps(cfg), gen.pps(cfg)]
for k in range(40):
    try:
        if k % 6 == 3:
            sps = fp.hostile_sps(base.sps[0], rnd)
            nals.append(b.nalu(3, 7, sps=sps))
            if rnd.randrange(2):
                nals.append(b.nalu(3, 8, sps=sps,
                                   pps=fp.hostile_pps(base.pps[0], rnd)))
        elif k % 6 == 0 and k:
            nals.append(b.nalu(3, 8, sps=base.sps[0],
                               pps=fp.hostile_pps(base.pps[0], rnd)))
    except OSError:
        pass
    st = rnd.choice([2, 0, 1, 7, 5, 6, 3, 4])
    first = rnd.choice([0, 0, rnd.randrange(total - 1)])
    n = rnd.randrange(1, total - first + 1)
    s = streams.rich_slice(cfg, seed * 991 + k, st, first, n,
                           density=rnd.randrange(0, 12),
                           hdr_extras=rnd.randrange(8) if st % 5 != 2 else 0)
    nal = gen.slice_nal(cfg, s)
    nals.append(mutate(nal, rnd) if k % 4 == 1 else nal)
flags = rnd.choice([0, 1, 1])
ref = orc.parse_nalus(nals, flags=flags, want=0)
got = prod.parse_nalus(nals, flags=flags)
print("flags", flags, "nals", len(nals), "statuses equal", got.statuses == ref.statuses)
a, c = got.event_tuples(), ref.event_tuples()
print("events", len(a), len(c))
for i in range(min(len(a), len(c))):
    if a[i] != c[i]:
        print("first difference at event", i)
        j = i
        while j > 0 and a[j][0] != 13:
            j -= 1
        print("  NAL event", a[j], "then", [e for e in a[j + 1:j + 8]])
        k = a[j][1]
        print("  NAL", k, "type", nals[k][0] & 31, "bytes", len(nals[k]),
              nals[k][:16].hex())
        # parameter sets in force: the last SPS / PPS NAL units before it
        for q in range(k - 1, -1, -1):
            if nals[q][0] & 31 in (7, 8):
                print("  earlier NAL", q, "type", nals[q][0] & 31,
                      nals[q][:24].hex())
            if k - q > 12:
                break
        for j in range(max(0, i - 4), min(len(a), i + 6)):
            print("  got ", j, a[j])
        for j in range(max(0, i - 4), min(len(c), i + 6)):
            print("  want", j, c[j])
        break
else:
    print("common prefix equal; tails:", a[len(c):][:5], c[len(a):][:5])
