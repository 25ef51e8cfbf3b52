#!/usr/bin/env python3
"""Hostile input through the DROP-IN API: mixed NAL units (parameter sets, SEI,
AUD, filler, slices of all types), most of them damaged, NAL by NAL through
h264_reader_parse_nalu and as an Annex-B stream through h264_reader_parse;
status codes and the complete callback log must equal the oracle's.

    python tools/fuzz_dropin.py [first_seed] [count] [params]

With `params`: SPS/PPS structures with extreme field values between the slices
(the lists of tools/fuzz_params.py).
"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import orc  # noqa: E402
import prod  # noqa: E402
import streams  # noqa: E402
from test_gpu_fuzz import mutate  # noqa: E402


def one(seed):
    rnd = random.Random(seed)
    cf, bd = rnd.choice([(1, 0), (1, 0), (0, 0), (2, 2), (3, 0)])
    # every fourth seed MBAFF material, every fourth slice groups (any map
    # type; frame, field or MBAFF pictures)
    mode = seed % 4
    if mode == 1:
        cfg = streams.small_cfg(cf=cf, bd=bd, field_coding=2)
        nals = streams.mixed_nals(cfg, n_slices=20, seed0=seed * 31)
    elif mode == 2:
        from libh264_amd import gen
        from test_oracle_vs_generator import FMO_CASES, fmo_cfg
        cfg, cycle = fmo_cfg(rnd.choice(list(FMO_CASES)), cf=cf, bd=bd,
                             field_coding=rnd.choice([0, 0, 1, 2]))
        nals = [gen.sps(cfg), gen.pps(cfg)]
        for pic in range(5):
            fp = 0 if not cfg.field_coding else rnd.choice([0, 1, 2])
            pair = cfg.field_coding == 2 and fp == 0
            if pic % 2 == 0:
                nals.append(gen.aud(pic % 3))
            for j, (first, n, _a) in enumerate(
                    gen.group_slices(cfg, cycle, fp, 1 if pair else 2)):
                kw = {"field_pic": fp} if cfg.field_coding else {}
                nals.append(gen.slice_nal(cfg, streams.rich_slice(
                    cfg, seed * 31 + 7 * pic + j, rnd.choice([2, 0, 1]),
                    first // 2 if pair else first, n,
                    slice_group_change_cycle=cycle, **kw)))
    else:
        cfg = streams.small_cfg(cf=cf, bd=bd)
        nals = streams.mixed_nals(cfg, n_slices=20, seed0=seed * 31)
    out = []
    for i, n in enumerate(nals):
        out.append(mutate(n, rnd) if (i >= 2 and rnd.random() < 0.6) else n)
    flags = rnd.choice([0, 1, 1])
    ref = orc.parse_nalus(out, flags=flags, want=0)
    got = prod.parse_nalus(out, flags=flags)
    if got.statuses != ref.statuses:
        k = next(i for i, (a, b) in enumerate(zip(got.statuses, ref.statuses))
                 if a != b)
        return "seed %d nal %d status %d != %d (%s)" % (
            seed, k, got.statuses[k], ref.statuses[k], out[k][:24].hex())
    if got.event_tuples() != ref.event_tuples():
        return "seed %d: callback logs differ (per NAL)" % seed
    if bytes(got.headers) != bytes(ref.headers):
        return "seed %d: slice headers differ" % seed
    data = streams.annexb(out, long_codes=[rnd.random() < 0.5 for _ in out])
    ref = orc.parse_stream(data, flags=flags, want=0)
    got = prod.parse_stream(data, flags=flags)
    want = [e for e in ref.event_tuples() if e[0] != 13]
    if got.consumed != ref.consumed or got.event_tuples() != want:
        return "seed %d: stream parse differs" % seed
    return None


def one_params(seed):
    """the NAL lists of tools/fuzz_params.py (hostile SPS/PPS structures
    between slices) through the drop-in reader"""
    import fuzz_params as fp
    from libh264_amd import gen
    rnd = random.Random(seed)
    cf, bd = rnd.choice([(1, 0), (1, 0), (0, 0), (2, 0), (3, 0), (1, 2)])
    cfg = streams.small_cfg(cf=cf, bd=bd,
                            transform_8x8_mode_flag=rnd.randrange(2),
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
    if got.statuses != ref.statuses:
        k = next(i for i, (a, c) in enumerate(zip(got.statuses, ref.statuses))
                 if a != c)
        return "params seed %d nal %d status %d != %d (%s)" % (
            seed, k, got.statuses[k], ref.statuses[k], nals[k][:24].hex())
    if got.event_tuples() != ref.event_tuples():
        return "params seed %d: callback logs differ" % seed
    if bytes(got.headers) != bytes(ref.headers):
        return "params seed %d: slice headers differ" % seed
    if bytes(got.sps) != bytes(ref.sps) or bytes(got.pps) != bytes(ref.pps):
        return "params seed %d: parameter sets differ" % seed
    return None


def main():
    import resource
    resource.setrlimit(resource.RLIMIT_AS, (96 << 30, 96 << 30))
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    params_mode = len(sys.argv) > 3 and sys.argv[3] == "params"
    if params_mode:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
    bad = 0
    for seed in range(first, first + count):
        msg = one_params(seed) if params_mode else one(seed)
        if msg:
            print(msg, flush=True)
            bad += 1
        if (seed - first) % 100 == 99:  # a sign of life for the job runner
            print("... %d seeds, %d divergences" % (seed - first + 1, bad),
                  flush=True)
    print("seeds %d..%d divergences: %d" % (first, first + count - 1, bad))


if __name__ == "__main__":
    main()
