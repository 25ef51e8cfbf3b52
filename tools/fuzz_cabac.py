#!/usr/bin/env python3
"""Volume fuzz of the CABAC subset kernel against the oracle's CPU statement of
the same decoder: slices from the independent Python encoder (random widths,
chroma formats, first_mb, QP, prediction / chroma modes, QP deltas), from the
writers' restatement (grey I, skipped P / B), opaque payloads -- most of them
then damaged.  Machinery of tests/test_gpu_cabac.py.

    python tools/fuzz_cabac.py [first_seed] [count]
"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import cabac_enc  # noqa: E402
import libh264_amd  # noqa: E402
import orc  # noqa: E402
from libh264_amd.ctypes_abi import TIER_T2, TIER_T1, TIER_T0  # noqa: E402
from test_cabac_subset import custom_i_slice, header, param_sets  # noqa: E402
from test_gpu_cabac import compare  # noqa: E402
from test_gpu_fuzz import mutate  # noqa: E402


def make_one(rnd, b, head, sps, pps, total, k):
    kind = rnd.randrange(4)
    first = rnd.choice([0, 0, rnd.randrange(total)]) if total > 1 else 0
    n = rnd.randrange(1, min(total - first, 150) + 1)
    qpd = rnd.randrange(-12, 12)
    if kind == 0 and not sps.frame_mbs_only_flag:
        kind = 1   # custom_i_slice builds frame-picture headers only
    if kind == 0:
        mbs = [(rnd.randrange(4), rnd.randrange(4),
                rnd.choice([0, 0, 0, 1, -1, 3, -7, 25, -26]))
               for _ in range(n)]
        return custom_i_slice(b, sps, pps, mbs, first, qpd, k)
    if kind == 1:
        sh = header(rnd.choice([5, 6, 0, 1, 8]), first, qpd,
                    rnd.randrange(3), k)
        if not sps.frame_mbs_only_flag:
            sh.field_pic_flag = rnd.randrange(2)
            sh.bottom_field_flag = rnd.randrange(2) & sh.field_pic_flag
        return b.nalu(2, 1, sps=sps, pps=pps, sh=sh, kind=2, mb_count=n,
                      cap=1 << 15)
    if kind == 2:
        sh = header(7, first, qpd, 0, k)
        return b.nalu(3, 5, sps=sps, pps=pps, sh=sh, kind=1, mb_count=n,
                      cap=1 << 15)
    st = rnd.choice([5, 6, 7])
    sh = header(st, first, qpd, rnd.randrange(3), k)
    hd = b.nalu(3 if st == 7 else 2, 5 if st == 7 else 1, sps=sps, pps=pps,
                sh=sh, kind=1 if st == 7 else 2, mb_count=1, cap=1 << 15)
    r = orc.parse_nalus(head + [hd], flags=0).results[0]
    if r.status != 0:
        return None
    return cabac_enc.escape_tail(
        hd[:r.rawdata_off],
        bytes(rnd.randrange(256) for _ in range(rnd.randrange(0, 80))))


def one(engine, seed):
    rnd = random.Random(seed)
    width = rnd.choice([1, 2, 5, 8, 11, 20, 45, 120])
    height = rnd.choice([1, 2, 6, 9])
    cf = rnd.choice([1, 1, 1, 2, 0, 3])
    sps, pps = param_sets(width, height, cf)
    if cf == 3:
        sps.profile_idc = 244
    if rnd.randrange(4) == 0:
        sps.frame_mbs_only_flag = 0      # field pictures / MBAFF frames
        sps.mb_adaptive_frame_field_flag = rnd.randrange(2)
    pps.pic_init_qp_minus26 = rnd.randrange(-26, 26)
    b = orc.Builder()
    head = [b.nalu(3, 7, sps=sps), b.nalu(3, 8, sps=sps, pps=pps)]
    total = width * height
    nals = []
    for k in range(24):
        try:
            nal = make_one(rnd, b, head, sps, pps, total, k)
        except OSError:
            continue  # the writer refused the combination
        if nal is not None:
            nals.append(mutate(nal, rnd) if rnd.randrange(3) else nal)
    if not nals:
        return None
    tier = rnd.choice([TIER_T2, TIER_T2, TIER_T1, TIER_T0])
    try:
        compare(engine, head, nals, tier=tier, cap=rnd.choice([total + 2, 7]),
                lead=seed % 4)
    except AssertionError as e:
        return "seed %d (w %d h %d cf %d): %s" % (seed, width, height, cf,
                                                 str(e)[:300])
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
