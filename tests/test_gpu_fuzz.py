"""GPU parity on hostile input, at volume: hundreds of slices per case with
random bit flips, truncations, byte insertions and spliced tails, decoded in
one batch and compared slice by slice with the oracle -- status, number of
macroblocks delivered before the error, checksum over their full syntax, the
records themselves.  Error paths are where a restructured kernel drifts from
the reference first (which error wins, how many macroblocks were already out)."""
import ctypes as C
import random

import numpy as np
import pytest

import orc
import streams
from libh264_amd import gen
from libh264_amd.ctypes_abi import SliceResult, TIER_T2, t2_stride
from test_gpu_decode import make_params, result_fields

pytestmark = pytest.mark.gpu


def mutate(nal, rnd):
    n = bytearray(nal)
    for _ in range(rnd.randrange(1, 4)):
        op = rnd.randrange(6)
        if op == 0 and len(n) > 6:       # flip one bit, anywhere behind the header
            p = rnd.randrange(1, len(n))
            n[p] ^= 1 << rnd.randrange(8)
        elif op == 1 and len(n) > 6:     # truncate
            del n[rnd.randrange(3, len(n)):]
        elif op == 2 and len(n) > 6:     # insert bytes rich in 00 / 03 / ff
            p = rnd.randrange(2, len(n))
            n[p:p] = bytes(rnd.choice([0, 0, 3, 1, 255])
                           for _ in range(rnd.randrange(1, 6)))
        elif op == 3 and len(n) > 12:    # zero a stretch (escape bytes, huge ue)
            p = rnd.randrange(2, len(n) - 4)
            for k in range(p, min(len(n), p + rnd.randrange(1, 9))):
                n[k] = 0
        elif op == 4 and len(n) > 12:    # all ones: long runs of empty blocks
            p = rnd.randrange(2, len(n) - 4)
            for k in range(p, min(len(n), p + rnd.randrange(1, 9))):
                n[k] = 0xff
        elif op == 5 and len(n) > 12:    # splice the tail somewhere else
            p = rnd.randrange(4, len(n) - 4)
            q = rnd.randrange(4, len(n) - 4)
            n[p:] = n[q:]
    return bytes(n)


@pytest.mark.parametrize("cf,bd,seed", [(1, 0, 1), (1, 0, 2), (0, 0, 3),
                                        (2, 2, 4), (3, 0, 5), (1, 0, 6),
                                        (1, 0, 7), (2, 0, 8)])
def test_damaged_slices_at_volume(engine, cf, bd, seed):
    rnd = random.Random(seed)
    # seeds 7 and 8: interlaced material with pic_order_cnt_type 1, SP and SI
    # slices among the types
    extra = dict(field_coding=1, pic_order_cnt_type=1,
                 bottom_field_pic_order_in_frame_present_flag=1) \
        if seed >= 7 else {}
    types = (2, 0, 1, 7, 5, 6, 3, 4) if seed >= 7 else (2, 0, 1, 7, 5, 6)
    cfg = streams.small_cfg(cf=cf, bd=bd, **extra)
    total = cfg.width_mbs * cfg.height_mbs
    nals = []
    for k in range(240):
        st = types[k % len(types)]
        first = rnd.choice([0, 0, 0, rnd.randrange(total - 1)])
        n = rnd.randrange(1, total - first + 1) if first else total
        s = streams.rich_slice(cfg, seed * 1000 + k, st, first, n,
                               density=rnd.randrange(0, 15),
                               pm_escape=rnd.choice([0, 20, 300]),
                               field_pic=k % 3 if extra else 0)
        nal = gen.slice_nal(cfg, s)
        nals.append(mutate(nal, rnd) if k % 8 else nal)  # every 8th intact
    ref = orc.parse_nalus([gen.sps(cfg), gen.pps(cfg)] + nals)
    params = make_params(ref)
    buf, offs, sizes = streams.pack_nals(nals, lead=seed % 4)
    d_buf = engine.upload(buf)
    descs = [(o, s, params.epoch) for o, s in zip(offs, sizes)]
    # a damaged mb_skip_run can ask for millions of macroblocks: the batch
    # gets room for four pictures and the kernel must stop at that with
    # H264_AMD_E_MB_OVERFLOW, the macroblocks up to there delivered
    cap = 4 * total
    out = engine.decode_slices(params, d_buf, descs, reader_flags=1,
                               tier=TIER_T2, mb_capacity=cap,
                               want_headers=True)
    res = engine.fetch_results(out)
    hdrs = engine.fetch_headers(out)
    n_err = 0
    k_ok = 0  # the oracle records a header when the `slice` callback fired
    for i in range(len(nals)):
        want, got = result_fields(ref.results[i]), result_fields(res[i])
        n_err += want["status"] != 0
        hdr_idx = k_ok
        k_ok += want["status"] == 0
        if want["mb_count"] > cap or (want["mb_count"] == cap and
                                      got["status"] == -75):
            assert got["status"] == -75 and got["mb_count"] == cap, (i, got)
            st = t2_stride(got["chroma_array_type"])
            mb = engine.fetch_mb(out, i, cap)[:, :st]
            assert np.array_equal(mb, ref.mb_records(i)[:cap]), i
            continue
        assert got == want, (i, got, want)
        if want["status"] == 0:
            assert bytes(hdrs[i]) == bytes(ref.headers[hdr_idx]), i
        if want["mb_count"]:
            st = t2_stride(want["chroma_array_type"])
            mb = engine.fetch_mb(out, i, want["mb_count"])[:, :st]
            assert np.array_equal(mb, ref.mb_records(i)), i
    assert 60 < n_err < 230  # the mutations do bite, and not always



@pytest.mark.parametrize("seed", [4070, 4226])
def test_dropin_fuzz_regressions(seed):
    """seeds of tools/fuzz_dropin.py that once diverged: a damaged
    mb_skip_run under slice groups, where the macroblock address runs ahead of
    the count -- oracle and kernel must end the slice at the same macroblock
    with the same status (H264_AMD_MAX_SLICE_MBS, include/h264_amd.h)"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(
        os.path.abspath(__file__))), "tools"))
    import fuzz_dropin
    assert fuzz_dropin.one(seed) is None
