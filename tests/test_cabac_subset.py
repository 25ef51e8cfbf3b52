"""CABAC decoder (engine extension, H264_AMD_READER_FLAGS_CABAC) on the subset of
the syntax the REFERENCE's writers emit -- the part of CABAC decoding that has
reference behaviour to pin against.  The oracle's CPU statement of it against (i) the reference writers' restatement
(round trip: what h264_write_grey_i_slice / h264_write_skipped_p_slice emit
must decode to exactly the macroblocks that were asked for) and (ii) an
independent Python CABAC encoder with values those writers never produce.
CPU only; the GPU kernel is compared with this oracle in test_gpu_cabac.py."""
import random

import numpy as np
import pytest

import cabac_enc
import orc
from libh264_amd.ctypes_abi import (Sps, Pps, SliceHeader,
                                    READER_FLAGS_CABAC_SUBSET)

FLAGS = 1 | READER_FLAGS_CABAC_SUBSET


def param_sets(width=120, height=68, cf=1):
    sps = Sps()
    sps.profile_idc = 100
    sps.level_idc = 40
    sps.chroma_format_idc = cf
    sps.pic_order_cnt_type = 2
    sps.max_num_ref_frames = 1
    sps.pic_width_in_mbs_minus1 = width - 1
    sps.pic_height_in_map_units_minus1 = height - 1
    sps.frame_mbs_only_flag = 1
    sps.direct_8x8_inference_flag = 1
    pps = Pps()
    pps.entropy_coding_mode_flag = 1
    pps.deblocking_filter_control_present_flag = 1
    return sps, pps


def header(st, first, qpd, idc, k=0):
    sh = SliceHeader()
    sh.slice_type = st
    sh.first_mb_in_slice = first
    sh.frame_num = k % 16
    sh.slice_qp_delta = qpd
    sh.cabac_init_idc = 0 if st % 5 == 2 else idc
    sh.idr_pic_id = k
    if st % 5 == 1:
        sh.direct_spatial_mv_pred_flag = 1
    return sh


GRID = [(8160, 0, 0, 0), (1, 0, 0, 0), (119, 1, -3, 1), (120, 0, 5, 2),
        (121, 7, 12, 0), (500, 4000, -20, 1), (3000, 120, 25, 2),
        (8160, 0, -26, 0), (64, 8096, 3, 2)]


def writer_slices(b, sps, pps):
    """(nal, expected mb_count, kind) from the writers' restatement."""
    out = []
    for k, (mbc, first, qpd, idc) in enumerate(GRID):
        for kind, st, nal_type, ref in ((0, 7, 5, 3), (1, 5, 1, 2),
                                        (1, 1, 1, 0)):
            sh = header(st, first, qpd, idc, k)
            out.append((b.nalu(ref, nal_type, sps=sps, pps=pps, sh=sh,
                               kind=kind + 1, mb_count=mbc, cap=1 << 16),
                        mbc, kind, st % 5))
    return out


def custom_i_slice(b, sps, pps, mbs, first, qpd, k=0):
    """A grey-I slice from the writer provides the header bytes; the slice
    data is replaced by the Python encoder's."""
    sh = header(7, first, qpd, 0, k)
    nal = b.nalu(3, 5, sps=sps, pps=pps, sh=sh, kind=1, mb_count=1,
                 cap=1 << 16)
    r = orc.parse_nalus([b.nalu(3, 7, sps=sps), b.nalu(3, 8, sps=sps, pps=pps),
                         nal], flags=0).results[0]
    qp = 26 + pps.pic_init_qp_minus26 + qpd
    payload = cabac_enc.encode_i16_slice(
        mbs, sps.pic_width_in_mbs_minus1 + 1, first, 0, qp,
        cat=sps.chroma_format_idc)
    return cabac_enc.escape_tail(nal[:r.rawdata_off], payload)


def test_python_encoder_equals_writer_restatement():
    """the test's own encoder reproduces the writers byte for byte"""
    sps, pps = param_sets()
    b = orc.Builder()
    head = [b.nalu(3, 7, sps=sps), b.nalu(3, 8, sps=sps, pps=pps)]
    for nal, mbc, kind, st in writer_slices(b, sps, pps)[:15]:
        r = orc.parse_nalus(head + [nal], flags=0).results[0]
        qp = r.slice_qp
        hdr = orc.parse_nalus(head + [nal], flags=0).headers[0]
        if kind == 0:
            pay = cabac_enc.encode_i16_slice([(2, 0, 0)] * mbc, 120,
                                             hdr.first_mb_in_slice, 0, qp)
        else:
            pay = cabac_enc.encode_skip_slice(mbc, hdr.slice_type,
                                              hdr.cabac_init_idc, qp)
        assert cabac_enc.escape_tail(nal[:r.rawdata_off], pay) == nal


def test_roundtrip_of_the_writers():
    sps, pps = param_sets()
    b = orc.Builder()
    sl = writer_slices(b, sps, pps)
    nals = [b.nalu(3, 7, sps=sps), b.nalu(3, 8, sps=sps, pps=pps)]
    ref = orc.parse_nalus(nals + [s[0] for s in sl], flags=FLAGS)
    off = orc.parse_nalus(nals + [s[0] for s in sl], flags=1)
    for i, (nal, mbc, kind, st) in enumerate(sl):
        r = ref.results[i]
        assert (r.status, r.mb_count) == (0, mbc), i
        assert r.flags == 7
        rec = ref.mb_records(i)
        if kind == 0:   # grey I: I_16x16_2_0_0, chroma DC, qp delta 0
            assert (rec[:, 0] == 2).all() and (rec[:, 3] == 3).all()
            assert (rec[:, 7] == 2).all() and not rec[:, 4:7].any()
        else:
            assert (rec[:, 0] == (16 if st == 1 else 10)).all()
            assert (rec[:, 2] == 1).all()
        assert not rec[:, 8:].any()
        # without the flag: the reference's behaviour, nothing delivered
        assert (off.results[i].mb_count, off.results[i].flags) == (0, 5)


@pytest.mark.parametrize("width,cf", [(120, 1), (8, 1), (11, 2), (5, 0)])
def test_values_the_writers_never_emit(width, cf):
    rnd = random.Random(width * 7 + cf)
    sps, pps = param_sets(width, 9, cf)
    b = orc.Builder()
    head = [b.nalu(3, 7, sps=sps), b.nalu(3, 8, sps=sps, pps=pps)]
    nals, want = [], []
    for k in range(12):
        first = rnd.choice([0, 0, rnd.randrange(width * 9 - 1)])
        n = rnd.randrange(1, width * 9 - first + 1)
        mbs = [(rnd.randrange(4), rnd.randrange(4),
                rnd.choice([0, 0, 0, 1, -1, 2, -3, 7, -26, 25]))
               for _ in range(n)]
        nals.append(custom_i_slice(b, sps, pps, mbs, first,
                                   rnd.randrange(-10, 10), k))
        want.append(mbs)
    ref = orc.parse_nalus(head + nals, flags=FLAGS)
    for i, mbs in enumerate(want):
        r = ref.results[i]
        assert (r.status, r.mb_count) == (0, len(mbs)), i
        rec = ref.mb_records(i)
        got = [(int(x[7]), int(x[6]), int(np.int16(x[4] | (x[5] << 8))))
               for x in rec[:, :8].astype(np.uint16)]
        exp = [(p, c if cf in (1, 2) else 0, q) for p, c, q in mbs]
        assert got == exp, i
        assert (rec[:, 3] == 1 + rec[:, 7]).all()


def test_unsupported_and_damaged_are_reported():
    """MBAFF frames with CABAC end with E_UNSUPPORTED, damaged data with -EIO,
    never with a wrong macroblock count."""
    import streams
    from libh264_amd import gen
    cfg = streams.small_cfg(entropy_coding_mode_flag=1, field_coding=2)
    nal = gen.slice_nal(cfg, streams.rich_slice(cfg, 1, 2, cabac_payload=40))
    r = orc.parse_nalus([gen.sps(cfg), gen.pps(cfg), nal],
                        flags=FLAGS).results[0]
    assert (r.status, r.mb_count, r.flags) == (-38, 0, 5)
    b = orc.Builder()
    sps, pps = param_sets(8, 6, 1)
    head = [b.nalu(3, 7, sps=sps), b.nalu(3, 8, sps=sps, pps=pps)]
    good = custom_i_slice(b, sps, pps, [(1, 2, -1)] * 48, 0, 0)
    r = orc.parse_nalus(head + [good], flags=FLAGS).results[0]
    assert (r.status, r.mb_count) == (0, 48)
    cut = good[:len(good) - 6]
    r = orc.parse_nalus(head + [cut], flags=FLAGS).results[0]
    assert r.status == -5 and r.mb_count < 48
