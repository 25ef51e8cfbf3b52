"""GPU parity for the Annex-B scan kernel, the slice write-back kernel and the
synthetic-slice / CABAC kernels, through the C-ABI, against the oracle."""
import ctypes as C
import random

import numpy as np
import pytest

import orc
import streams
from libh264_amd import gen, Params
from libh264_amd.ctypes_abi import (Sps, Pps, SliceHeader, TIER_NONE, TIER_T2)
from test_gpu_decode import make_params

pytestmark = pytest.mark.gpu


def oracle_nals(data):
    rec = orc.parse_stream(data, flags=0, want=0)
    return [(n.start, n.end) for n in rec.nals], rec.consumed


def gpu_nals(engine, data, cap=None):
    buf = np.frombuffer(bytes(data) + b"\xff" * 64, dtype=np.uint8)
    d_buf = engine.upload(buf)
    if cap is None:
        cap = len(data) // 3 + 8
    tab, count, consumed = engine.scan(d_buf, len(data), cap)
    return [(int(a), int(b)) for a, b in tab], count, consumed


def check_scan(engine, data):
    want, want_off = oracle_nals(data)
    got, count, consumed = gpu_nals(engine, data)
    assert count == len(want)
    assert got == want
    assert consumed == want_off


def test_scan_hand_cases(engine):
    cases = [
        b"\x00\x00\x01\x65\x88",
        b"\x00\x00\x00\x01\x65\x88\x00\x00\x01\x41\x9a",
        b"\xab\xcd\x00\x00\x01\x09\x10\x00\x00\x00\x01\x67\x42\x00\x00\x00",
        b"\x00\x00\x01\x00\x00\x01\x00\x00\x01",        # empty NAL units
        b"\x00\x00\x00\x00\x00\x00\x00\x00\x01\x65",    # long zero run
        b"\x00\x00\x01\x65\x00\x00\x03\x00\x00\x03\x01",  # escapes inside
        b"\x00\x00\x01\x65\x00\x00",                    # ends in two zeros
        b"\x00\x00\x01",                                # nothing behind
        b"\x01\x02\x03\x04",                            # no start code
        b"\x00\x00",                                    # too short
        b"\x00\x00\x01\x65" + b"\x00" * 40 + b"\x01\x41",
    ]
    for c in cases:
        check_scan(engine, c)


@pytest.mark.parametrize("seed", range(6))
def test_scan_random_dense(engine, seed):
    """Byte soup rich in 00/01/03 so that patterns straddle every boundary."""
    rnd = random.Random(seed)
    n = rnd.choice([100, 1023, 1024, 1025, 16383, 16384, 16385, 70001])
    data = bytes(rnd.choice([0, 0, 0, 1, 3, 7, 255]) for _ in range(n))
    check_scan(engine, data)


def test_scan_segment_boundaries(engine):
    """Start codes placed around the 16 KiB segment and 1 KiB step edges."""
    for edge in (1024, 4096, 16384, 32768):
        for shift in range(-5, 3):
            data = bytearray(b"\x55" * (edge + 3000))
            data[10:13] = b"\x00\x00\x01"
            p = edge + shift
            data[p:p + 4] = b"\x00\x00\x00\x01"
            data[p + 900:p + 903] = b"\x00\x00\x01"
            check_scan(engine, bytes(data))


@pytest.mark.parametrize("seg", [4096, 65536])
def test_scan_other_segment_sizes(seg, monkeypatch):
    """Large buffers are walked in longer per-wave segments (engine picks the
    size from the buffer length); H264_AMD_SCAN_SEG forces one so that small
    inputs reach the same code: many tiny segments, and segments holding
    several pipelined 4 KiB blocks."""
    import libh264_amd
    monkeypatch.setenv("H264_AMD_SCAN_SEG", str(seg))
    eng = libh264_amd.Engine(0)
    rnd = random.Random(seg)
    # sparse codes in zero-free payload (the fast filter's normal diet) ...
    data = bytearray(rnd.randrange(1, 256) for _ in range(300000))
    for p in sorted(rnd.sample(range(0, len(data) - 8), 60)):
        data[p:p + 4] = rnd.choice([b"\x00\x00\x01\x65", b"\x00\x00\x00\x01",
                                    b"\x00\x00\x00\x00", b"\x00\x00\x03\x00"])
    # ... codes straddling every kind of edge ...
    for edge in (1024, 4096, 8192, seg, 2 * seg, 3 * seg + 1024):
        for shift in (-3, -2, -1, 0, 1):
            p = edge + shift + 16 * (shift + 3)
            if p + 4 < len(data):
                data[p:p + 3] = b"\x00\x00\x01"
    check_scan(eng, bytes(data))
    # ... and byte soup
    soup = bytes(rnd.choice([0, 0, 0, 1, 3, 7, 255]) for _ in range(70001))
    check_scan(eng, soup)


def test_scan_mixed_stream(engine):
    cfg = streams.small_cfg()
    nals = [gen.sps(cfg), gen.pps(cfg), gen.aud(0), gen.sei_user_data(300, 3)]
    for seed in range(40):
        nals.append(gen.slice_nal(cfg, streams.rich_slice(cfg, seed,
                                                          (2, 0, 1)[seed % 3])))
        if seed % 7 == 0:
            nals.append(gen.filler(50))
    rnd = random.Random(5)
    data = b"\x00" * 3 + streams.annexb(
        nals, long_codes=[rnd.random() < 0.5 for _ in nals]) + b"\x00\x00"
    check_scan(engine, data)


def test_gather_nal_heads(engine):
    """first bytes of every NAL unit in one table; short units zero padded"""
    cfg = streams.small_cfg()
    nals = [gen.sps(cfg), gen.pps(cfg), gen.aud(0), b"\x09", b""] + \
        [gen.slice_nal(cfg, streams.rich_slice(cfg, s, 2)) for s in range(5)]
    data = streams.annexb([n for n in nals if n])
    buf = np.frombuffer(data + b"\xff" * 64, dtype=np.uint8)
    d_buf = engine.upload(buf)
    tab, count, consumed, d_nals = engine.scan(d_buf, len(data), 64, keep=True)
    for k in (1, 7, 16, 64, 256):
        heads = engine.gather_nal_heads(d_buf, d_nals, count, k)
        assert heads.shape == (count, k)
        for i in range(count):
            a, b = int(tab[i, 0]), int(tab[i, 1])
            want = data[a:min(b, a + k)]
            assert heads[i, :len(want)].tobytes() == want
            assert not heads[i, len(want):].any()
    import ctypes as C
    from libh264_amd import capi
    L = capi.engine_lib()
    assert L.h264_amd_gather_nal_heads(engine.h, d_buf.ptr, d_nals.ptr, 3, 0,
                                       d_buf.ptr) == -22
    assert L.h264_amd_gather_nal_heads(engine.h, d_buf.ptr, d_nals.ptr, 3, 257,
                                       d_buf.ptr) == -22
    assert L.h264_amd_gather_nal_heads(engine.h, None, d_nals.ptr, 3, 8,
                                       d_buf.ptr) == -22
    assert L.h264_amd_gather_nal_heads(engine.h, d_buf.ptr, d_nals.ptr, 0, 8,
                                       None) == 0


def test_scan_cap_too_small(engine):
    data = b"".join(b"\x00\x00\x01\x09\x10" for _ in range(100))
    got, count, _ = gpu_nals(engine, data, cap=10)
    assert count == 100


def roundtrip_case(engine, cfg, slices, lead=0):
    sps, pps = gen.sps(cfg), gen.pps(cfg)
    nals = [gen.slice_nal(cfg, s) for s in slices]
    ref = orc.parse_nalus([sps, pps] + nals, flags=0,
                          want=orc.ORC_REC_WANT_REWRITE)
    params = make_params(ref)
    buf, offs, sizes = streams.pack_nals(nals, lead=lead)
    d_buf = engine.upload(buf)
    descs = [(o, s, params.epoch) for o, s in zip(offs, sizes)]
    out = engine.decode_slices(params, d_buf, descs, reader_flags=0,
                               tier=TIER_NONE, want_headers=True)
    cap = max(sizes) + 64
    d_out, ooffs, osizes = engine.write_slices(params, d_buf, out, cap)
    raw = engine.download(d_out)
    rew = {w.nal_index: w for w in ref.rewrite_idx}
    for i, n in enumerate(nals):
        assert osizes[i] == len(n), (i, osizes[i], len(n))
        got = raw[int(ooffs[i]):int(ooffs[i]) + len(n)].tobytes()
        assert got == n, i
        w = rew[i + 2]
        assert w.match == 1 and w.len == len(n)


def test_write_roundtrip_420(engine):
    cfg = streams.small_cfg()
    slices = [streams.rich_slice(cfg, seed, st, first_mb=seed % 5)
              for seed in range(24) for st in (2, 0, 1)]
    roundtrip_case(engine, cfg, slices)


@pytest.mark.parametrize("lead", [1, 2, 3])
def test_write_roundtrip_unaligned(engine, lead):
    cfg = streams.small_cfg(weighted_pred_flag=1, weighted_bipred_idc=1)
    slices = [streams.rich_slice(cfg, 50 + seed, st)
              for seed in range(8) for st in (2, 0, 1)]
    roundtrip_case(engine, cfg, slices, lead=lead)


def test_write_roundtrip_1080p(engine):
    cfg = gen.cfg_1080p()
    slices = [streams.rich_slice(cfg, 3 + k, 7, pm_pcm=2, density=5,
                                 pm_escape=5, hdr_extras=0) for k in range(4)]
    roundtrip_case(engine, cfg, slices)


def _ps(entropy, t8=0):
    sps = Sps()
    sps.profile_idc = 100
    sps.level_idc = 40
    sps.chroma_format_idc = 1
    sps.pic_order_cnt_type = 2
    sps.max_num_ref_frames = 1
    sps.pic_width_in_mbs_minus1 = 119
    sps.pic_height_in_map_units_minus1 = 67
    sps.frame_mbs_only_flag = 1
    sps.direct_8x8_inference_flag = 1
    pps = Pps()
    pps.entropy_coding_mode_flag = entropy
    pps.deblocking_filter_control_present_flag = 1
    return sps, pps


@pytest.mark.parametrize("entropy", [0, 1])
def test_synth_slices(engine, entropy):
    """h264_write_grey_i_slice / h264_write_skipped_p_slice on the GPU vs the
    oracle's restatement, over (mb_count, first_mb, QP, cabac_init_idc)."""
    sps, pps = _ps(entropy)
    params = Params()
    params.put_sps(sps)
    epoch = params.put_pps(pps)
    b = orc.Builder()
    descs, hdrs, want = [], [], []
    for k, (mbc, first, qpd, idc) in enumerate(
            [(8160, 0, 0, 0), (1, 0, 0, 0), (119, 1, -3, 1), (120, 0, 5, 2),
             (121, 7, 12, 0), (500, 4000, -20, 1), (3000, 120, 25, 2),
             (8160, 0, -26, 0), (64, 8096, 3, 3)]):
        for kind, st, nal_type, ref in ((0, 7, 5, 3), (1, 5, 1, 2),
                                        (1, 1, 1, 0)):
            sh = SliceHeader()
            sh.slice_type = st
            sh.first_mb_in_slice = first
            sh.frame_num = k % 16
            sh.slice_qp_delta = qpd
            sh.cabac_init_idc = idc
            sh.idr_pic_id = k
            if st % 5 == 1:
                sh.direct_spatial_mv_pred_flag = 1
            if st % 5 == 2 or entropy == 0:
                sh.cabac_init_idc = 0
            descs.append((kind, epoch, mbc, ref, nal_type))
            hdrs.append(sh)
            want.append(b.nalu(ref, nal_type, sps=sps, pps=pps, sh=sh,
                               kind=kind + 1, mb_count=mbc, cap=1 << 16))
    got = engine.encode_synth(params, descs, hdrs, 1 << 14)
    for i, (g, w) in enumerate(zip(got, want)):
        assert g == w, (i, descs[i], len(w))
    # sizes recorded from a run of the real reference
    if entropy == 0:
        assert len(got[0]) == 8164
    else:
        assert len(got[0]) == 402
        assert len(got[1]) == 63


def test_cabac_init_states(engine):
    triples = [(st, idc, qp) for st in (0, 1, 2, 3, 4) for idc in (0, 1, 2, 3)
               for qp in range(-5, 60)]
    got = engine.cabac_init_states(triples)
    for i, (st, idc, qp) in enumerate(triples):
        want = orc.cabac_init_states(st, idc, qp)
        assert np.array_equal(got[i], want), (st, idc, qp)


def test_byte_stream_to_avcc(engine):
    """h264_amd_byte_stream_to_avcc (the scan in four-byte-code mode + length
    writer) against the oracle's restatement of the reference's loop, in
    place: hand cases, byte soup, codes around step and segment edges, a real
    stream, and a stream dense enough to need the second (exact) table pass."""
    import ctypes as C
    from test_host_control_plane import _avcc_cases
    O = orc.lib()
    O.orc_byte_stream_to_avcc.argtypes = [C.c_void_p, C.c_size_t]
    cases = [c for c in _avcc_cases() if len(c) > 0]
    rnd = random.Random(9)
    for edge in (1024, 4096, 16384, 32768):
        for shift in range(-6, 3):
            d = bytearray(rnd.randrange(1, 256) for _ in range(edge + 2000))
            d[3:7] = b"\x00\x00\x00\x01"
            p = edge + shift
            d[p:p + 5] = b"\x00\x00\x00\x00\x01"
            d[p + 700:p + 703] = b"\x00\x00\x01"      # three bytes: payload
            cases.append(bytes(d))
    dense = bytearray(rnd.randrange(1, 256) for _ in range(120000))
    for p in range(0, len(dense) - 8, 40):
        dense[p:p + 4] = b"\x00\x00\x00\x01"
    cases.append(bytes(dense))
    big = bytearray(np.random.default_rng(5).integers(1, 256, size=5 << 20,
                                                      dtype=np.uint8).tobytes())
    for p in sorted(rnd.sample(range(0, len(big) - 8), 300)):
        big[p:p + 4] = b"\x00\x00\x00\x01"
    cases.append(bytes(big))
    for data in cases:
        want = C.create_string_buffer(data, len(data))
        assert O.orc_byte_stream_to_avcc(want, len(data)) == 0
        d_buf = engine.upload(np.frombuffer(data + b"\xff" * 64, dtype=np.uint8))
        engine.byte_stream_to_avcc(d_buf, len(data))
        got = engine.download(d_buf, len(data)).tobytes()
        assert got == want.raw, (len(data), data[:12])
        assert engine.download(d_buf, 64, offset=len(data)).tobytes() == \
            b"\xff" * 64


@pytest.mark.parametrize("case", [3, 4, 5, 1])
def test_write_roundtrip_slice_groups_and_mbaff(engine, case):
    """write-back of slices under slice groups (slice_group_change_cycle is
    re-encoded with the width the picture size and the change rate give it,
    ref src/h264_syntax.h:1352-1372) and of MBAFF frames: byte-identical"""
    from test_oracle_vs_generator import fmo_cfg, fmo_slices, mbaff_slices
    cfg, cycle = fmo_cfg(case)
    roundtrip_case(engine, cfg, [s for s, _ in fmo_slices(cfg, cycle, 2300)])
    cfg, cycle = fmo_cfg(case, field_coding=1)
    roundtrip_case(engine, cfg, [s for s, _ in
                                 fmo_slices(cfg, cycle, 2400, field_pic=1)])
    cfg = streams.small_cfg(field_coding=2)
    roundtrip_case(engine, cfg, mbaff_slices(cfg, 2500, n=6))
