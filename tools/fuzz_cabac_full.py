#!/usr/bin/env python3
"""Volume fuzz of the CABAC slice-data kernel (h264_amd_k_decode_cabac) against
the oracle's decoder on REAL content: slices from the generator's own arithmetic
encoder -- I / SI / P / SP / B, four chroma formats, bit depths, 8x8 transform,
I_PCM, field pictures, partial slices, every cabac_init_idc, the QP range
(below 0 too, with either clip of the context initialisation), wide
and narrow pictures -- two in three of them then damaged (bit flips, truncation,
spliced tails, padding).  Status, macroblocks delivered before an error, header
images and records in a random tier must be equal.  Machinery of
tests/test_gpu_cabac.py.

    python tools/fuzz_cabac_full.py [first_seed] [count]
"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import libh264_amd  # noqa: E402
import streams  # noqa: E402
from libh264_amd import gen  # noqa: E402
from libh264_amd.ctypes_abi import TIER_T2, TIER_T1, TIER_T0, TIER_NONE  # noqa: E402
from test_cabac_full import FLAGS, STD_QP, cabac_cfg, cabac_slice  # noqa: E402
from test_gpu_cabac import compare  # noqa: E402
from test_gpu_fuzz import mutate  # noqa: E402


def one(engine, seed):
    rnd = random.Random(seed)
    # either QP clip of the context initialisation, in the slices and in the
    # decoders (a stream of its own: earlier seeds keep their slices)
    rq = random.Random(seed ^ 0x51c0)
    flags = FLAGS | (STD_QP if rq.randrange(2) else 0)
    cf = rnd.choice([1, 1, 1, 2, 0, 3])
    bd = rnd.choice([0, 0, 2, 4]) if cf != 1 else rnd.choice([0, 0, 0, 2])
    width = rnd.choice([1, 2, 3, 5, 8, 11, 20, 45])
    height = rnd.choice([1, 2, 3, 6])
    fc = rnd.choice([0, 0, 0, 1])
    cfg = cabac_cfg(cf=cf, bd=bd, width_mbs=width, height_mbs=height,
                    field_coding=fc,
                    transform_8x8_mode_flag=rnd.randrange(4) != 0,
                    direct_8x8_inference_flag=rnd.randrange(2) or fc,
                    pic_init_qp_minus26=rnd.randrange(-20, 20),
                    num_ref_idx_l0_default_minus1=rnd.randrange(3),
                    num_ref_idx_l1_default_minus1=rnd.randrange(2))
    head = [gen.sps(cfg), gen.pps(cfg)]
    total = width * height
    nals = []
    for k in range(20):
        first = rnd.choice([0, 0, rnd.randrange(total)]) if total > 1 else 0
        n = rnd.randrange(1, total - first + 1)
        s = cabac_slice(cfg, seed * 131 + k, rnd.choice([2, 0, 1, 3, 4, 7, 5]),
                        first, n, cabac_init_idc=rnd.randrange(3),
                        slice_qp_delta=rnd.randrange(-12, 12),
                        density=rnd.randrange(0, 15),
                        pm_pcm=rnd.choice([0, 30, 300]),
                        pm_zero_pcm=rnd.choice([0, 900]),
                        pm_escape=rnd.choice([0, 20, 400]),
                        pm_t8x8=rnd.choice([0, 400, 900]),
                        pm_skip=rnd.choice([0, 200, 700]),
                        pm_cbp_zero=rnd.choice([0, 100, 600]),
                        field_pic=rnd.randrange(3) if fc else 0,
                        cabac_syntax=rq.choice([1, 1, 2]))
        try:
            nal = gen.slice_nal(cfg, s)
        except OSError:
            continue
        r = rnd.randrange(3)
        if r == 0:
            nals.append(nal)
        elif r == 1:
            nals.append(mutate(nal, rnd))
        else:
            b = bytearray(nal)
            for _ in range(rnd.randrange(1, 4)):
                p = rnd.randrange(min(4, len(b) - 1), len(b))
                b[p] ^= 1 << rnd.randrange(8)
            nals.append(bytes(b))
    if not nals:
        return None
    tier = rnd.choice([TIER_T2, TIER_T2, TIER_T1, TIER_T0, TIER_NONE])
    try:
        compare(engine, head, nals, tier=tier,
                cap=rnd.choice([total + 2, total + 2, 7]), lead=seed % 4,
                flags=flags)
    except AssertionError as e:
        return "seed %d (w %d h %d cf %d bd %d fc %d): %s" % (
            seed, width, height, cf, bd, fc, str(e)[:int(os.environ.get("FUZZ_MSG", "300"))])
    return None


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    eng = libh264_amd.Engine(0)
    bad = 0
    for seed in range(first, first + count):
        msg = one(eng, seed)
        if msg:
            print(msg, flush=True)
            bad += 1
        if (seed - first) % 500 == 499:
            print("... %d seeds, %d divergences" % (seed - first + 1, bad),
                  flush=True)
    print("seeds %d..%d divergences: %d" % (first, first + count - 1, bad))


if __name__ == "__main__":
    main()
