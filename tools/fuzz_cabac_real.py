#!/usr/bin/env python3
"""Volume fuzz of the CABAC kernel against the oracle on damaged REAL slices
(the two third-party clips of tests/test_cabac_real_streams.py, where the image
has them): bit flips, truncations, splices, padding; status, macroblocks
delivered before an error and checksums must be equal.

    python tools/fuzz_cabac_real.py [first_seed] [count]
"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import libh264_amd  # noqa: E402
from libh264_amd.ctypes_abi import TIER_T0, TIER_T1, TIER_T2, TIER_NONE  # noqa: E402
from test_cabac_real_streams import quirk, slices_of  # noqa: E402
from test_gpu_cabac import compare  # noqa: E402
from test_gpu_fuzz import mutate  # noqa: E402
import glob  # noqa: E402


def clip(name):
    c = glob.glob("/opt/conda/lib/python3*/site-packages/imageio/resources/"
                  "images/" + name)
    if not c:
        raise SystemExit("no real clips in this image")
    return slices_of(c[0])


def load_clips():
    return [clip("cockatoo.mp4") + (3602,), clip("realshort.mp4") + (302,)]


def one(eng, clips, seed):
    """one seed: raises AssertionError on a divergence"""
    rnd = random.Random(seed)
    head, nals, cap = clips[rnd.randrange(2)]
    pick = [rnd.choice(nals) for _ in range(24)]
    batch = [n if rnd.randrange(4) == 0 else mutate(n, rnd) for n in pick]
    q = rnd.randrange(2) == 1
    flags = 0x101 | (0x200 if q else 0)
    tier = rnd.choice([TIER_T0, TIER_T0, TIER_T1, TIER_NONE] +
                      ([TIER_T2] if cap < 1000 else []))
    try:
        quirk(q)
        compare(eng, head, batch, tier=tier, cap=rnd.choice([cap, cap, 50]),
                flags=flags, lead=seed % 4)
    finally:
        quirk(False)


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    eng = libh264_amd.Engine(0)
    clips = load_clips()
    bad = 0
    for seed in range(first, first + count):
        try:
            one(eng, clips, seed)
        except AssertionError as e:
            print("seed %d: %s" % (seed, str(e)[:int(os.environ.get("FUZZ_MSG", "300"))]), flush=True)
            bad += 1
        if (seed - first) % 50 == 49:
            print("... %d seeds, %d divergences" % (seed - first + 1, bad),
                  flush=True)
    print("seeds %d..%d divergences: %d" % (first, first + count - 1, bad))


if __name__ == "__main__":
    main()
