#!/usr/bin/env python3
"""Longer hunt with the machinery of tests/test_gpu_fuzz.py: many seeds,
all chroma formats, header extras on.  Prints the first divergence per seed.

    python tools/fuzz_campaign.py [first_seed] [count]
"""
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import libh264_amd  # noqa: E402
import orc  # noqa: E402
import streams  # noqa: E402
from libh264_amd import gen  # noqa: E402
from libh264_amd.ctypes_abi import TIER_T2, t2_stride  # noqa: E402
from test_gpu_decode import make_params, result_fields  # noqa: E402
from test_gpu_fuzz import mutate  # noqa: E402


def one(engine, seed):
    rnd = random.Random(seed)
    cf, bd = rnd.choice([(1, 0), (1, 0), (0, 0), (2, 0), (2, 2), (3, 0), (3, 4)])
    cfg = streams.small_cfg(cf=cf, bd=bd,
                            transform_8x8_mode_flag=rnd.randrange(2),
                            weighted_pred_flag=rnd.randrange(2),
                            weighted_bipred_idc=rnd.randrange(3))
    # every fourth seed: MBAFF frames; every fourth: slice groups (a random map
    # type, frame / field / MBAFF pictures); the rest as before
    mode = ("plain", "mbaff", "fmo", "plain")[seed % 4]
    cycle = 0
    if mode == "mbaff":
        cfg = streams.small_cfg(cf=cf, bd=bd, field_coding=2,
                                transform_8x8_mode_flag=rnd.randrange(2),
                                weighted_pred_flag=rnd.randrange(2),
                                weighted_bipred_idc=rnd.randrange(3))
    elif mode == "fmo":
        from test_oracle_vs_generator import FMO_CASES, fmo_cfg
        cfg, cycle = fmo_cfg(rnd.choice(list(FMO_CASES)), cf=cf, bd=bd,
                             field_coding=rnd.choice([0, 0, 1, 2]),
                             transform_8x8_mode_flag=rnd.randrange(2))
    total = cfg.width_mbs * cfg.height_mbs
    nals = []
    groups = {}
    for k in range(160):
        st = rnd.choice([2, 0, 1, 7, 5, 6, 3, 4])
        kw = {}
        fp = rnd.randrange(3) if cfg.field_coding == 1 else 0
        if cfg.field_coding == 2:
            fp = rnd.choice([0, 0, 1, 2])
        if cfg.field_coding:
            kw["field_pic"] = fp
        pair = cfg.field_coding == 2 and fp == 0
        if mode == "fmo":
            # a slice of one group: from one of its addresses to some later one
            if fp not in groups:
                groups[fp] = gen.slice_group_map(cfg, cycle, fp)
            m = groups[fp]
            addrs = np.nonzero(m == rnd.choice(sorted(set(m.tolist()))))[0]
            step = 2 if pair else 1
            i0 = rnd.choice([0, rnd.randrange(len(addrs) // step) * step])
            first = int(addrs[i0]) // step
            n = rnd.randrange(1, (len(addrs) - i0) // step + 1) * step
            kw["slice_group_change_cycle"] = cycle
        else:
            units = total  # pairs of an MBAFF frame, macroblocks otherwise
            first = rnd.choice([0, 0, rnd.randrange(units - 1)])
            n = rnd.randrange(1, units - first + 1) * (2 if pair else 1)
        s = streams.rich_slice(cfg, seed * 977 + k, st, first, n,
                               density=rnd.randrange(0, 17),
                               pm_escape=rnd.choice([0, 30, 400]),
                               pm_pcm=rnd.choice([0, 30, 300]),
                               pm_skip=rnd.choice([200, 200, 700]),
                               hdr_extras=rnd.randrange(8) if st % 5 != 2 else 0,
                               **kw)
        nal = gen.slice_nal(cfg, s)
        nals.append(mutate(nal, rnd) if k % 6 else nal)
    ref = orc.parse_nalus([gen.sps(cfg), gen.pps(cfg)] + nals)
    params = make_params(ref)
    buf, offs, sizes = streams.pack_nals(nals, lead=seed % 5)
    d_buf = engine.upload(buf)
    descs = [(o, s, params.epoch) for o, s in zip(offs, sizes)]
    cap = 8 * total
    out = engine.decode_slices(params, d_buf, descs, reader_flags=1,
                               tier=TIER_T2, mb_capacity=cap,
                               want_headers=True)
    res = engine.fetch_results(out)
    hdrs = engine.fetch_headers(out)
    k_ok = 0
    for i in range(len(nals)):
        want, got = result_fields(ref.results[i]), result_fields(res[i])
        hdr_idx = k_ok
        k_ok += want["status"] == 0
        if want["mb_count"] > cap or (want["mb_count"] == cap and
                                      got["status"] == -75):
            if not (got["status"] == -75 and got["mb_count"] == cap):
                return "seed %d slice %d overflow mismatch %r" % (seed, i, got)
            continue
        if got != want:
            diff = {k: (got[k], want[k]) for k in got if got[k] != want[k]}
            return "seed %d (cf %d bd %d) slice %d: %r  nal=%s" % (
                seed, cf, bd, i, diff, nals[i][:48].hex())
        if want["status"] == 0 and bytes(hdrs[i]) != bytes(ref.headers[hdr_idx]):
            return "seed %d slice %d header image differs" % (seed, i)
        if want["mb_count"]:
            st = t2_stride(want["chroma_array_type"])
            mb = engine.fetch_mb(out, i, want["mb_count"])[:, :st]
            if not np.array_equal(mb, ref.mb_records(i)):
                return "seed %d slice %d records differ" % (seed, i)
    return None


def main():
    # never let a run-away input take the whole box down: cap our own address
    # space well below the machine's memory
    import resource
    resource.setrlimit(resource.RLIMIT_AS, (96 << 30, 96 << 30))
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    eng = libh264_amd.Engine(0)
    bad = 0
    for seed in range(first, first + count):
        msg = one(eng, seed)
        print(seed, "ok" if msg is None else msg, flush=True)
        bad += msg is not None
    print("divergences:", bad)


if __name__ == "__main__":
    main()
