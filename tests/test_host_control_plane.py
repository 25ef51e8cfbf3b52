"""Control plane of the drop-in library on the CPU: everything that is NOT a
slice (parameter sets with VUI/HRD/scaling lists, AUD, filler, every SEI
payload type, stream info, AVCC helpers, Annex-B single-shot search) runs on
the host by design and is compared here with the oracle.  Slice entry points
must refuse to run without a GPU (-ENODEV) instead of falling back."""
import ctypes as C
import os
import random
import subprocess

import pytest

import orc
import prod
import streams
from h264types import Bitstream, NaluHeader, SliceHeader
from libh264_amd import gen

ROOT = orc.ROOT
_CTL = None


def ctl():
    global _CTL
    if _CTL is None:
        orc.lib()
        prod.h264()
        src = os.path.join(ROOT, "tests", "helpers", "ctlplane.c")
        out_dir = os.path.join(ROOT, "tests", "helpers", "build")
        out = os.path.join(out_dir, "libctlplane.so")
        os.makedirs(out_dir, exist_ok=True)
        odir = os.path.join(ROOT, "oracle", "build")
        subprocess.check_call([
            "gcc", "-std=gnu99", "-O1", "-g", "-fPIC", "-shared", "-Wall",
            "-I" + os.path.join(ROOT, "include"), src, "-o", out,
            "-L" + prod.LIBDIR, "-lh264", "-L" + odir, "-loracle",
            "-Wl,-rpath," + prod.LIBDIR, "-Wl,-rpath," + odir])
        L = C.CDLL(out)
        L.ctl_crosscheck.argtypes = [C.c_uint64]
        L.ctl_make_stream.argtypes = [C.c_uint64, C.c_void_p, C.c_size_t]
        L.ctl_avcc_roundtrip.argtypes = [C.c_void_p, C.c_size_t]
        _CTL = L
    return _CTL


def test_write_parse_crosscheck_random_structures():
    L = ctl()
    bad = [(seed, L.ctl_crosscheck(seed)) for seed in range(1, 401)]
    bad = [b for b in bad if b[1] != 0]
    assert bad == []


def test_sei_streams_parse_identically():
    """SPS + PPS + SEI NAL units written by the oracle from random structures,
    parsed NAL by NAL by both readers: same callbacks, same typed payloads."""
    L = ctl()
    buf = C.create_string_buffer(1 << 17)
    kinds = set()
    for seed in range(1, 121):
        n = L.ctl_make_stream(seed, buf, len(buf))
        assert n > 0, seed
        nals = [x for x in buf.raw[:n].split(b"\x00\x00\x00\x01") if x]
        assert len(nals) == 3
        ref = orc.parse_nalus(nals, want=0)
        got = prod.parse_nalus(nals)
        assert got.statuses == ref.statuses, seed
        assert got.event_tuples() == ref.event_tuples(), seed
        assert bytes(got.sps) == bytes(ref.sps)
        assert bytes(got.pps) == bytes(ref.pps)
        kinds |= {e[1] for e in got.event_tuples() if e[0] == 12}
    assert kinds == {0, 1, 2, 3, 4, 5, 6}  # every typed SEI callback fired


def test_non_slice_nals_with_damage():
    cfg = streams.small_cfg(vui=1)
    base = [gen.sps(cfg), gen.pps(cfg), gen.aud(2), gen.sei_user_data(90, 5),
            gen.filler(33), gen.sei_user_data(3, 1), b"\x0e\x11\x22",
            b"\x14\x80"]
    rnd = random.Random(2)
    nals = list(base)
    for _ in range(200):
        n = bytearray(rnd.choice(base))
        for _k in range(rnd.randrange(1, 4)):
            op = rnd.randrange(3)
            if op == 0 and len(n) > 1:
                del n[rnd.randrange(1, len(n)):]
            elif op == 1:
                n[rnd.randrange(len(n))] ^= 1 << rnd.randrange(8)
            else:
                n += bytes(rnd.randrange(256) for _ in range(rnd.randrange(4)))
        if (n[0] & 0x1f) in (1, 5) and not n[0] & 0x80:
            continue  # a slice: GPU test territory
        nals.append(bytes(n))
    ref = orc.parse_nalus(nals, want=0)
    got = prod.parse_nalus(nals)
    assert got.statuses == ref.statuses
    assert got.event_tuples() == ref.event_tuples()
    assert bytes(got.sps) == bytes(ref.sps)
    assert bytes(got.pps) == bytes(ref.pps)
    assert any(s < 0 for s in got.statuses)


def test_find_nalu_matches_oracle():
    L = prod.h264()
    rnd = random.Random(4)
    cases = [b"", b"\x00", b"\x00\x00\x01", b"\x00\x00\x01\x65",
             b"\x00\x00\x00\x01\x67\x00\x00\x01\x68", b"\x01\x02\x03",
             b"\x00\x00\x01\x65\x00\x00", b"\x00\x00\x01\x00\x00\x00"]
    cases += [bytes(rnd.choice([0, 0, 0, 1, 2, 3, 9]) for _ in
                    range(rnd.randrange(0, 60))) for _ in range(400)]
    for data in cases:
        s, e = C.c_size_t(7), C.c_size_t(7)
        res = L.h264_find_nalu(data, len(data), C.byref(s), C.byref(e))
        want = orc.find_nalu(data)
        assert res == want[0], data
        if res in (0, -11):
            assert (s.value, e.value) == want[1:], data


def test_avcc_roundtrip():
    L = ctl()
    cfg = streams.small_cfg()
    nals = [gen.sps(cfg), gen.pps(cfg), gen.aud(0),
            gen.slice_nal(cfg, streams.rich_slice(cfg, 1, 2))]
    data = streams.annexb(nals)
    buf = C.create_string_buffer(data, len(data))
    assert L.ctl_avcc_roundtrip(buf, len(data)) == 0
    H = prod.h264()
    H.h264_byte_stream_to_avcc.argtypes = [C.c_void_p, C.c_size_t]
    assert H.h264_byte_stream_to_avcc(buf, len(data)) == 0
    pos = 0
    for n in nals:
        assert int.from_bytes(buf.raw[pos:pos + 4], "big") == len(n)
        assert buf.raw[pos + 4:pos + 4 + len(n)] == n
        pos += 4 + len(n)


def has_gpu():
    L = C.CDLL(os.path.join(prod.LIBDIR, "libh264_amd_engine.so"))
    return bool(L.h264_amd_available())


@pytest.mark.skipif(has_gpu(), reason="a GPU is present")
def test_slice_paths_fail_loudly_without_gpu():
    """No CPU slice path: every slice entry point reports -ENODEV."""
    cfg = streams.small_cfg()
    sl = gen.slice_nal(cfg, streams.rich_slice(cfg, 1, 2))
    got = prod.parse_nalus([gen.sps(cfg), gen.pps(cfg), sl])
    assert got.statuses == [0, 0, -19]
    st = prod.parse_stream(streams.annexb([gen.sps(cfg), sl]))
    assert st.result == -19
    L = prod.h264()
    ctx = C.c_void_p()
    L.h264_ctx_new(C.byref(ctx))
    from h264types import Sps, Pps
    sps, pps = Sps(), Pps()
    s, p = gen.sps(cfg), gen.pps(cfg)
    assert L.h264_parse_sps(s, len(s), C.byref(sps)) == 0
    assert L.h264_parse_pps(p, len(p), C.byref(sps), C.byref(pps)) == 0
    L.h264_ctx_set_sps(ctx, C.byref(sps))
    L.h264_ctx_set_pps(ctx, C.byref(pps))
    nh = NaluHeader(0, 1, 5)
    L.h264_ctx_set_nalu_header(ctx, C.byref(nh))
    sh = SliceHeader()
    sh.slice_type = 7
    L.h264_ctx_set_slice_header(ctx, C.byref(sh))
    out = C.create_string_buffer(4096)
    for fn, args in ((L.h264_write_nalu, ()),
                     (L.h264_write_grey_i_slice, (10,)),
                     (L.h264_write_skipped_p_slice, (10,))):
        bs = Bitstream()
        bs.data = C.cast(out, C.c_void_p).value
        bs.len = len(out)
        bs.emulation_prevention = 1
        assert fn(C.byref(bs), ctx, *args) == -19, fn
        assert bs.off == 0
    L.h264_ctx_destroy(ctx)


def test_find_nalu_argument_checks():
    """SURVEY Appendix C.15 (ref src/h264_bitstream.c:163-165)."""
    L = prod.h264()
    s, e = C.c_size_t(), C.c_size_t()
    data = b"\x00\x00\x01\x65"
    L.h264_find_nalu.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p,
                                 C.c_void_p]
    assert L.h264_find_nalu(None, 4, C.byref(s), C.byref(e)) == -22
    assert L.h264_find_nalu(data, 4, None, C.byref(e)) == -22
    assert L.h264_find_nalu(data, 4, C.byref(s), None) == -22
    # -EAGAIN: start found, no end marker; -ENOENT: no start code at all
    assert L.h264_find_nalu(data, 4, C.byref(s), C.byref(e)) == -11
    assert (s.value, e.value) == (3, 4)
    assert L.h264_find_nalu(b"\x01\x02\x03", 3, C.byref(s), C.byref(e)) == -2


def test_sei_pic_timing_clock_conversion():
    """h264_ctx_sei_pic_timing_to_ts / _to_us (ref src/h264_ctx.c:491-542):
    ((h * 60 + m) * 60 + s) * time_scale + n_frames * num_units_in_tick *
    (1 + nuit_field_based_flag) + time_offset, clamped at 0; us rounded."""
    L = prod.h264()
    from h264types import Sps

    class Clk(C.Structure):
        _fields_ = [(n, C.c_int32) for n in (
            "clock_timestamp_flag ct_type nuit_field_based_flag counting_type "
            "full_timestamp_flag discontinuity_flag cnt_dropped_flag n_frames "
            "seconds_value minutes_value hours_value seconds_flag minutes_flag "
            "hours_flag time_offset").split()]

    class PicTiming(C.Structure):
        _fields_ = [("cpb_removal_delay", C.c_uint32),
                    ("dpb_output_delay", C.c_uint32),
                    ("pic_struct", C.c_uint32), ("clk_ts", Clk * 3)]

    assert C.sizeof(PicTiming) == 192
    for f in (L.h264_ctx_sei_pic_timing_to_ts, L.h264_ctx_sei_pic_timing_to_us):
        f.restype = C.c_uint64
        f.argtypes = [C.c_void_p, C.POINTER(PicTiming)]
    ctx = C.c_void_p()
    L.h264_ctx_new(C.byref(ctx))
    sps = Sps()
    sps.vui_parameters_present_flag = 1
    sps.vui.timing_info_present_flag = 1
    sps.vui.num_units_in_tick = 1001
    sps.vui.time_scale = 60000
    L.h264_ctx_set_sps(ctx, C.byref(sps))
    pt = PicTiming()
    c = pt.clk_ts[0]
    c.hours_value, c.minutes_value, c.seconds_value = 1, 2, 3
    c.n_frames, c.nuit_field_based_flag, c.time_offset = 7, 1, -5
    want = ((1 * 60 + 2) * 60 + 3) * 60000 + 7 * 1001 * 2 - 5
    assert L.h264_ctx_sei_pic_timing_to_ts(ctx, C.byref(pt)) == want
    assert L.h264_ctx_sei_pic_timing_to_us(ctx, C.byref(pt)) == \
        (want * 1000000 + 30000) // 60000
    c.hours_value = c.minutes_value = c.seconds_value = c.n_frames = 0
    c.time_offset = -9  # would go negative: clamped to 0
    assert L.h264_ctx_sei_pic_timing_to_ts(ctx, C.byref(pt)) == 0
    L.h264_ctx_destroy(ctx)


def _avcc_cases():
    import random
    rnd = random.Random(4)
    cases = [b"\x00\x00\x00\x01", b"\x00\x00\x00\x01\x65", b"abc",
             b"\x00\x00\x01\x65\x00\x00\x00\x01\x41\x9a",
             b"\x00\x00\x00\x01\x00\x00\x00\x01\x09",
             b"\x11\x00\x00\x00\x00\x01\x67\x42\x00\x00\x00\x01",
             b"\x00\x00\x00\x00\x00\x01\x67" + b"\x55" * 70000 +
             b"\x00\x00\x00\x00\x01\x68\x00"]
    for n in (5, 64, 1000, 20000):
        cases.append(bytes(rnd.choice([0, 0, 0, 1, 1, 3, 0x65, 0xff])
                           for _ in range(n)))
    cfg = streams.small_cfg()
    nals = [gen.sps(cfg), gen.pps(cfg), gen.aud(0)] + \
        [gen.slice_nal(cfg, streams.rich_slice(cfg, s, 2)) for s in range(6)]
    cases.append(streams.annexb(nals, long_codes=[True] * len(nals)))
    cases.append(streams.annexb(nals, long_codes=[i % 2 == 0 for i in
                                                  range(len(nals))]) + b"\x00")
    return cases


def test_avcc_against_the_oracle():
    """h264_byte_stream_to_avcc / h264_avcc_to_byte_stream of the drop-in
    library vs the oracle's restatement of src/h264.c:184-272: mixed 3- and
    4-byte start codes, zero runs, empty units, codes at the very end, byte
    soup; then hostile AVCC buffers in the other direction."""
    import orc
    H = prod.h264()
    O = orc.lib()
    for f in (H.h264_byte_stream_to_avcc, H.h264_avcc_to_byte_stream,
              O.orc_byte_stream_to_avcc, O.orc_avcc_to_byte_stream):
        f.argtypes = [C.c_void_p, C.c_size_t]
    for data in _avcc_cases():
        a = C.create_string_buffer(data, len(data))
        b = C.create_string_buffer(data, len(data))
        ra = H.h264_byte_stream_to_avcc(a, len(data))
        rb = O.orc_byte_stream_to_avcc(b, len(data))
        assert ra == rb and a.raw == b.raw, data[:16]
        # and back, on whatever came out (a valid AVCC buffer only when the
        # input started with a four-byte code and had no other bytes first)
        ra = H.h264_avcc_to_byte_stream(a, len(data))
        rb = O.orc_avcc_to_byte_stream(b, len(data))
        assert ra == rb and a.raw == b.raw, data[:16]
    # a zero length is -EPROTO; a length running past the end is accepted (the
    # reference does not look); fewer than four bytes left for a length is
    # -EPROTO here, where the reference reads past the buffer
    for bad, want in ((b"\x00\x00\x00\x00", -71),
                      (b"\x00\x00\x00\x02\x65", 0),
                      (b"\x00\x00\x00\x01\x65\x00", -71),
                      (b"\x00\x00\x00\x01\x65\x00\x00\x00", -71)):
        a = C.create_string_buffer(bad, len(bad))
        b = C.create_string_buffer(bad, len(bad))
        assert H.h264_avcc_to_byte_stream(a, len(bad)) == want
        assert O.orc_avcc_to_byte_stream(b, len(bad)) == want
        assert a.raw == b.raw


def test_avcc_random_buffers_against_the_oracle():
    """2000 random buffers rich in start-code fragments through both
    directions, product host code vs oracle, in place"""
    import random
    import orc
    H = prod.h264()
    O = orc.lib()
    for f in (H.h264_byte_stream_to_avcc, H.h264_avcc_to_byte_stream,
              O.orc_byte_stream_to_avcc, O.orc_avcc_to_byte_stream):
        f.argtypes = [C.c_void_p, C.c_size_t]
    rnd = random.Random(77)
    frags = [b"\x00\x00\x00\x01", b"\x00\x00\x01", b"\x00", b"\x00\x00",
             b"\x00\x00\x00", b"\x01", b"\x65\x88", b"\xff", b"\x00\x00\x00\x02"]
    for _ in range(2000):
        data = b"".join(rnd.choice(frags) if rnd.random() < 0.6 else
                        bytes(rnd.randrange(256) for _ in range(rnd.randrange(1, 9)))
                        for _ in range(rnd.randrange(1, 30)))
        for fh, fo in ((H.h264_byte_stream_to_avcc, O.orc_byte_stream_to_avcc),
                       (H.h264_avcc_to_byte_stream, O.orc_avcc_to_byte_stream)):
            a = C.create_string_buffer(data, len(data) + 8)
            b = C.create_string_buffer(data, len(data) + 8)
            assert fh(a, len(data)) == fo(b, len(data)), data.hex()
            assert a.raw == b.raw, data.hex()
            assert a.raw[len(data):] == b"\x00" * 8  # nothing past the end
