"""GPU parity of the DROP-IN library (libh264_amd/lib/libh264.so): the
reference's own public API -- h264_reader_parse / h264_reader_parse_nalu with
callbacks, h264_write_nalu, h264_write_grey_i_slice,
h264_write_skipped_p_slice, h264_rewrite_slice_header -- driven through
tests/helpers/recorder.c and compared event for event, byte for byte, with the
oracle's restatement of the same entry points (ref src/h264_reader.c:113-162,
src/h264_syntax.h:1446-1604, src/h264_writer.c:254-370)."""
import ctypes as C
import random

import numpy as np
import pytest

import orc
import prod
import streams
from libh264_amd import gen
from libh264_amd.ctypes_abi import (Bitstream, NaluHeader, SliceHeader, Sps,
                                    Pps)

pytestmark = pytest.mark.gpu

EV_PARSE_RESULT = 13


def mixed_nals(cfg, n_slices=24, seed0=0, split=True):
    """SPS, PPS, then pictures made of AUD + SEI + one or two slices, with a
    filler now and then and a repeated parameter-set pair in the middle."""
    nals = [gen.sps(cfg), gen.pps(cfg)]
    total = cfg.width_mbs * cfg.height_mbs
    for k in range(n_slices):
        seed = seed0 + k
        st = (2, 0, 1, 0)[k % 4]
        if k % 3 == 0:
            nals.append(gen.aud(k % 3))
        if k % 5 == 0:
            nals.append(gen.sei_user_data(40 + k, seed + 1))
        if split and k % 2:
            half = total // 2
            nals.append(gen.slice_nal(
                cfg, streams.rich_slice(cfg, seed, st, 0, half)))
            nals.append(gen.slice_nal(
                cfg, streams.rich_slice(cfg, seed, st, half, total - half)))
        else:
            nals.append(gen.slice_nal(cfg, streams.rich_slice(cfg, seed, st)))
        if k % 7 == 3:
            nals.append(gen.filler(20 + k))
        if k == n_slices // 2:
            nals += [gen.sps(cfg), gen.pps(cfg)]
    return nals


def same_log(ref, got, stream):
    want = ref.event_tuples()
    have = got.event_tuples()
    if stream:
        # h264_reader_parse swallows the per-NAL status
        want = [e for e in want if e[0] != EV_PARSE_RESULT]
    assert len(have) == len(want)
    for i, (w, h) in enumerate(zip(want, have)):
        assert w == h, (i, w, h)
    assert bytes(got.sps) == bytes(ref.sps)
    assert bytes(got.pps) == bytes(ref.pps)
    assert bytes(got.headers) == bytes(ref.headers)


@pytest.mark.parametrize("flags", [0, 1])
def test_reader_stream(flags):
    cfg = streams.small_cfg()
    nals = mixed_nals(cfg)
    rnd = random.Random(3)
    data = b"\x00\x00" + streams.annexb(
        nals, long_codes=[rnd.random() < 0.5 for _ in nals]) + b"\x00"
    ref = orc.parse_stream(data, flags=flags, want=0)
    got = prod.parse_stream(data, flags=flags)
    assert got.result == 0
    assert got.consumed == ref.consumed
    same_log(ref, got, stream=True)
    kinds = {e[0] for e in got.event_tuples()}
    assert {1, 2, 3, 4, 8, 9, 10, 11}.issubset(kinds)
    if flags:
        assert {5, 6, 7}.issubset(kinds)


@pytest.mark.parametrize("cf,bd", [(0, 0), (2, 2), (3, 0)])
def test_reader_stream_formats(cf, bd):
    cfg = streams.small_cfg(cf=cf, bd=bd)
    data = streams.annexb(mixed_nals(cfg, n_slices=8, seed0=50 + cf))
    ref = orc.parse_stream(data, want=0)
    got = prod.parse_stream(data)
    assert got.consumed == ref.consumed
    same_log(ref, got, stream=True)


@pytest.mark.parametrize("kw", [
    dict(field_coding=1, bottom_field_pic_order_in_frame_present_flag=1),
    dict(field_coding=1, pic_order_cnt_type=1),
    dict(pic_order_cnt_type=1, delta_pic_order_always_zero_flag=1),
    dict(field_coding=2),     # MBAFF frame pictures between field pictures
])
def test_reader_stream_interlaced_and_poc1(kw):
    """field pictures (access-unit detection looks at field_pic_flag and
    bottom_field_flag, 7.4.1.2.4; every macroblock callback of a field picture
    belongs to a field macroblock), pic_order_cnt_type 1, SP / SI slices"""
    cfg = streams.small_cfg(**kw)
    nals = streams.mixed_nals(cfg, n_slices=20, seed0=70,
                              types=(2, 0, 1, 3, 4))
    data = streams.annexb(nals)
    ref = orc.parse_stream(data, want=0)
    got = prod.parse_stream(data)
    assert got.consumed == ref.consumed
    same_log(ref, got, stream=True)
    ref = orc.parse_nalus(nals, want=0)
    got = prod.parse_nalus(nals)
    assert got.statuses == ref.statuses
    assert -38 not in ref.statuses
    same_log(ref, got, stream=False)


def test_reader_nalu_by_nalu_with_damage():
    """h264_reader_parse_nalu per NAL, statuses included; some NAL units are
    truncated, bit-flipped, empty, or carry the forbidden bit."""
    cfg = streams.small_cfg()
    nals = [bytearray(n) for n in mixed_nals(cfg, n_slices=16, seed0=9)]
    rnd = random.Random(11)
    for i in range(2, len(nals)):
        r = rnd.random()
        if r < 0.15 and len(nals[i]) > 8:
            del nals[i][rnd.randrange(4, len(nals[i])):]
        elif r < 0.30 and len(nals[i]) > 8:
            p = rnd.randrange(1, len(nals[i]))
            nals[i][p] ^= 1 << rnd.randrange(8)
        elif r < 0.34:
            nals[i][0] |= 0x80
    nals.insert(5, bytearray(b"\x65"))      # slice NAL with nothing behind
    nals.insert(9, bytearray(b"\x0c\xff\x80"))
    nals.insert(12, bytearray(b"\x18\x01\x02"))  # unknown type 24
    ref = orc.parse_nalus(nals, want=0)
    got = prod.parse_nalus(nals)
    assert got.statuses == ref.statuses
    same_log(ref, got, stream=False)
    assert any(s < 0 for s in got.statuses)


def test_reader_missing_parameter_sets():
    cfg = streams.small_cfg()
    sl = gen.slice_nal(cfg, streams.rich_slice(cfg, 1, 2))
    seq = [sl, gen.pps(cfg), sl, gen.sps(cfg), sl, gen.pps(cfg), sl]
    ref = orc.parse_nalus(seq, want=0)
    got = prod.parse_nalus(seq)
    assert got.statuses == ref.statuses
    assert got.statuses[0] < 0 and got.statuses[-1] == 0
    same_log(ref, got, stream=False)


def test_reader_parameter_set_change_mid_stream():
    """A second SPS/PPS pair with another picture size and entropy mode: each
    slice must bind to the versions current at ITS position."""
    a = streams.small_cfg()
    b = streams.small_cfg(width_mbs=5, height_mbs=4)
    nals = [gen.sps(a), gen.pps(a)]
    nals += [gen.slice_nal(a, streams.rich_slice(a, s, (2, 0)[s % 2]))
             for s in range(4)]
    nals += [gen.sps(b), gen.pps(b)]
    nals += [gen.slice_nal(b, streams.rich_slice(b, s, (2, 0, 1)[s % 3]))
             for s in range(4, 10)]
    nals += [gen.sps(a), gen.pps(a)]
    nals += [gen.slice_nal(a, streams.rich_slice(a, 20, 2))]
    data = streams.annexb(nals)
    ref = orc.parse_stream(data, want=0)
    got = prod.parse_stream(data)
    same_log(ref, got, stream=True)
    assert all(r.status == 0 for r in ref.results)


def test_reader_many_slices_chunks(monkeypatch):
    """1080p slices with the host image limited to two slices per chunk."""
    monkeypatch.setenv("H264_AMD_READER_CHUNK_BYTES", "100000")
    cfg = gen.cfg_1080p()
    sl = [gen.GenSlice(slice_type=7 if k % 3 == 0 else 5, idr=int(k % 3 == 0),
                       nal_ref_idc=1, first_mb=0, num_mbs=8160,
                       frame_num=k % 16, density=2, pm_skip=300, pm_i16x16=200,
                       seed=k + 1) for k in range(6)]
    nals = [gen.sps(cfg), gen.pps(cfg)] + [gen.slice_nal(cfg, s) for s in sl]
    data = streams.annexb(nals * 1)
    ref = orc.parse_stream(data, want=0)
    got = prod.parse_stream(data)
    same_log(ref, got, stream=True)
    assert sum(1 for e in got.event_tuples() if e[0] == 7) == 6 * 8160


@pytest.mark.parametrize("stop_after", [1, 2, 5, 9, 30])
def test_reader_stop_and_resume(monkeypatch, stop_after):
    """h264_reader_stop() from a nalu_end callback in the middle of a buffer
    (ref src/h264_reader.c:107-114,133): the callbacks up to there, the offset
    returned, and the rest of the buffer parsed with the same reader from that
    offset -- with record pages of a few slices, so that a page is on its way
    from the device when the parse is left."""
    monkeypatch.setenv("H264_AMD_READER_CHUNK_BYTES", "30000")
    cfg = streams.small_cfg()
    nals = mixed_nals(cfg, n_slices=30, seed0=50)
    data = streams.annexb(nals)
    ref = orc.parse_stream(data, want=0, stop_after=stop_after)
    got = prod.parse_stream(data, stop_after=stop_after)
    assert got.result == 0
    assert 0 < ref.consumed_first < len(data)
    assert got.consumed_first == ref.consumed_first
    assert got.consumed == ref.consumed
    same_log(ref, got, stream=True)


def test_dump_style_write_back():
    """The reference tool's check (tools/h264_dump.c:229-262): every NAL is
    re-written from nalu_end with h264_write_nalu and must reproduce its
    input.  Slices go through the write kernel, the rest through the host."""
    cfg = streams.small_cfg()
    nals = mixed_nals(cfg, n_slices=12, seed0=30)
    data = streams.annexb(nals)
    ref = orc.parse_stream(data, flags=0, want=orc.ORC_REC_WANT_REWRITE)
    got = prod.parse_stream(data, flags=0, want=prod.WANT_REWRITE)
    same_log(ref, got, stream=True)
    assert got.rewrites == ref.rewrites
    assert len(got.rewrite_idx) == len(ref.rewrite_idx) > 0
    for w, h in zip(ref.rewrite_idx, got.rewrite_idx):
        assert (w.off, w.len, w.res, w.match) == (h.off, h.len, h.res,
                                                  h.match)
        assert h.match == 1


# ---- writer entry points ---------------------------------------------------

class Ctx:
    """A product h264_ctx set up like an application would: parameter sets,
    NAL header, slice header."""

    def __init__(self, sps, pps):
        L = prod.h264()
        self.p = C.c_void_p()
        assert L.h264_ctx_new(C.byref(self.p)) == 0
        assert L.h264_ctx_set_sps(self.p, C.byref(sps)) == 0
        assert L.h264_ctx_set_pps(self.p, C.byref(pps)) == 0

    def slice(self, ref_idc, nal_type, sh):
        L = prod.h264()
        L.h264_ctx_clear_nalu(self.p)
        nh = NaluHeader(0, ref_idc, nal_type)
        assert L.h264_ctx_set_nalu_header(self.p, C.byref(nh)) == 0
        assert L.h264_ctx_set_slice_header(self.p, C.byref(sh)) == 0

    def close(self):
        prod.h264().h264_ctx_destroy(self.p)


def new_bs(prefix=b"", emulation=1):
    buf = C.create_string_buffer(1 << 16)
    bs = Bitstream()
    bs.data = C.cast(buf, C.c_void_p).value
    bs.len = len(buf)
    bs.emulation_prevention = emulation
    C.memmove(buf, prefix, len(prefix))
    bs.off = len(prefix)
    return bs, buf


def parsed_sets(cfg):
    sps, pps = Sps(), Pps()
    s, p = gen.sps(cfg), gen.pps(cfg)
    L = prod.h264()
    assert L.h264_parse_sps(s, len(s), C.byref(sps)) == 0
    assert L.h264_parse_pps(p, len(p), C.byref(sps), C.byref(pps)) == 0
    return sps, pps


@pytest.mark.parametrize("entropy", [0, 1])
@pytest.mark.parametrize("prefix", [b"", b"\x00\x00\x00\x01", b"\x00\x00"])
def test_synthetic_slices_public_api(entropy, prefix):
    cfg = gen.cfg_1080p(profile_idc=77 if entropy else 66,
                        entropy_coding_mode_flag=entropy)
    sps, pps = parsed_sets(cfg)
    ctx = Ctx(sps, pps)
    bld = orc.Builder()
    L = prod.h264()
    cases = [(1, 5, 7, 8160, 1), (1, 1, 5, 8160, 2), (0, 1, 5, 1, 2),
             (1, 5, 2, 77, 1), (1, 1, 0, 300, 2)]
    for ref_idc, nal_type, st, mbs, kind in cases:
        sh = SliceHeader()
        sh.slice_type = st
        sh.frame_num = 3
        sh.slice_qp_delta = -2
        if nal_type == 5:
            sh.idr_pic_id = 9
        want = bld.nalu(ref_idc, nal_type, sps=sps, pps=pps, sh=sh, kind=kind,
                        mb_count=mbs)
        ctx.slice(ref_idc, nal_type, sh)
        bs, buf = new_bs(prefix)
        fn = (L.h264_write_grey_i_slice if kind == 1 else
              L.h264_write_skipped_p_slice)
        assert fn(C.byref(bs), ctx.p, mbs) == 0
        got = buf.raw[len(prefix):bs.off]
        if prefix.endswith(b"\x00\x00") and want[0] <= 3:
            # the bytes already in the caller's bitstream take part in the
            # escape rule (ref src/h264_bitstream.c:54-81): a NAL header
            # byte 0x01 (nal_ref_idc 0, type 1) behind 00 00 is escaped
            want = b"\x03" + want
        assert got == want, (kind, st, mbs, len(got), len(want))
        assert bs.cachebits == 0
    ctx.close()


def test_grey_i_known_sizes():
    """SURVEY.md C2 known answers: 1080p grey IDR = 8164 B (CAVLC) / 402 B
    (CABAC); skipped P (CABAC) = 63 B."""
    L = prod.h264()
    for entropy, grey, skip in ((0, 8164, None), (1, 402, 63)):
        cfg = gen.cfg_1080p(profile_idc=77 if entropy else 66,
                            entropy_coding_mode_flag=entropy)
        sps, pps = parsed_sets(cfg)
        ctx = Ctx(sps, pps)
        sh = SliceHeader()
        sh.slice_type = 7
        ctx.slice(3, 5, sh)
        bs, buf = new_bs()
        assert L.h264_write_grey_i_slice(C.byref(bs), ctx.p, 8160) == 0
        want = orc.Builder().nalu(3, 5, sps=sps, pps=pps, sh=sh, kind=1,
                                  mb_count=8160)
        assert buf.raw[:bs.off] == want
        if skip is not None:
            sh = SliceHeader()
            sh.slice_type = 5
            sh.frame_num = 1
            ctx.slice(2, 1, sh)
            bs, buf = new_bs()
            assert L.h264_write_skipped_p_slice(C.byref(bs), ctx.p, 8160) == 0
            want = orc.Builder().nalu(2, 1, sps=sps, pps=pps, sh=sh, kind=2,
                                      mb_count=8160)
            assert buf.raw[:bs.off] == want
        ctx.close()


def test_write_nalu_slice_header_only_and_rewrite():
    """h264_write_nalu on a slice without raw data leaves the header's last
    bits pending in the bitstream cache; h264_rewrite_slice_header patches a
    same-length header in place and refuses another length with -EPROTO."""
    cfg = streams.small_cfg()
    sps, pps = parsed_sets(cfg)
    L = prod.h264()
    nal = gen.slice_nal(cfg, streams.rich_slice(cfg, 4, 0, hdr_extras=0))
    ref = orc.parse_nalus([gen.sps(cfg), gen.pps(cfg), nal], flags=0, want=0)
    sh = SliceHeader.from_buffer_copy(bytes(ref.headers)[:C.sizeof(SliceHeader)])
    r = ref.results[0]
    ctx = Ctx(sps, pps)
    ctx.slice(r.nal_ref_idc, r.nal_unit_type, sh)
    bs, buf = new_bs(b"\x00\x00\x01")
    assert L.h264_write_nalu(C.byref(bs), ctx.p) == 0
    bits = 8 * (bs.off - 3) + bs.cachebits
    assert bits == r.hdr_len_bits
    whole = bs.off - 3
    assert buf.raw[3:bs.off] == nal[:whole]
    if bs.cachebits:
        assert bs.cache == nal[whole] & (0xff00 >> bs.cachebits) & 0xff

    # in-place patch of the NAL: frame_num is u(v), same length
    patched = C.create_string_buffer(bytes(nal), len(nal))
    pbs = Bitstream()
    pbs.data = C.cast(patched, C.c_void_p).value
    pbs.len = len(nal)
    pbs.emulation_prevention = 1
    sh2 = SliceHeader.from_buffer_copy(bytes(sh))
    sh2.frame_num = (sh.frame_num + 5) % 16
    # ctx->slice.hdr_len counted from the NAL start, as a reader leaves it:
    # write the original header once into a scratch bitstream
    ctx.slice(r.nal_ref_idc, r.nal_unit_type, sh)
    sbs, _sbuf = new_bs()
    assert L.h264_write_nalu(C.byref(sbs), ctx.p) == 0
    assert L.h264_rewrite_slice_header(C.byref(pbs), ctx.p,
                                       C.byref(sh2)) == 0
    out = orc.parse_nalus([gen.sps(cfg), gen.pps(cfg), patched.raw], want=0)
    assert out.statuses[2] == 0
    got = SliceHeader.from_buffer_copy(
        bytes(out.headers)[:C.sizeof(SliceHeader)])
    assert got.frame_num == sh2.frame_num
    assert got.first_mb_in_slice == sh.first_mb_in_slice
    full = orc.parse_nalus([gen.sps(cfg), gen.pps(cfg), nal], want=0)
    assert out.results[0].checksum == full.results[0].checksum != 0

    # another length: ue(v) field grows
    sh3 = SliceHeader.from_buffer_copy(bytes(sh2))
    sh3.first_mb_in_slice = 40
    before = patched.raw
    assert L.h264_rewrite_slice_header(C.byref(pbs), ctx.p,
                                       C.byref(sh3)) == -71  # -EPROTO
    assert patched.raw == before
    ctx.close()


def test_ctx_clear_then_other_parameter_sets():
    """h264_ctx_clear() followed by as many set_sps / set_pps calls as came
    before it must not leave the reader's (or the writer's) device snapshot of
    the parameter sets in place: the generation counter they compare never
    repeats (libh264_amd/csrc/host/ctx.c)."""
    L, H = prod.recorder(), prod.h264()
    a = streams.small_cfg()
    b = streams.small_cfg(width_mbs=5, height_mbs=4, chroma_format_idc=2)
    first = [gen.sps(a), gen.pps(a),
             gen.slice_nal(a, streams.rich_slice(a, 3, 2))]
    second = [gen.sps(b), gen.pps(b),
              gen.slice_nal(b, streams.rich_slice(b, 4, 2)),
              gen.slice_nal(b, streams.rich_slice(b, 5, 0))]
    rec = L.rec_new(1, 0)
    keep = []

    def feed(nal):
        buf = C.create_string_buffer(bytes(nal), len(nal))
        keep.append(buf)
        return L.rec_parse_nalu(rec, C.addressof(buf), C.addressof(buf),
                                len(nal))

    assert [feed(n) for n in first] == [0, 0, 0]
    before = prod.Recording(rec, keep).event_tuples()
    n_before = len(before)
    # events 4 / 8 / 9 / 13 carry an index into the recording's arrays
    base = {k: sum(1 for e in before if e[0] == k) for k in (4, 8, 9, 13)}
    assert H.h264_ctx_clear(L.rec_ctx(rec)) == 0
    assert [feed(n) for n in second] == [0, 0, 0, 0]
    got = prod.Recording(rec, keep)
    L.rec_free(rec)
    ref = orc.parse_nalus(second, want=0)
    assert ref.statuses == [0, 0, 0, 0]
    tail = [(k, a - base.get(k, 0), b, c)
            for k, a, b, c in got.event_tuples()[n_before:]]
    assert tail == ref.event_tuples()
    assert bytes(got.headers)[-2 * C.sizeof(SliceHeader):] == \
        bytes(ref.headers)
    assert bytes(got.sps)[-C.sizeof(Sps):] == bytes(ref.sps)
    assert bytes(got.pps)[-C.sizeof(Pps):] == bytes(ref.pps)

    # the writer side: same sequence through a plain h264_ctx; a header
    # written under B must be the header of the NAL generated under B (frame
    # number width, chroma format and picture size all differ from A)
    sa, pa = parsed_sets(a)
    sb, pb = parsed_sets(b)
    nal = second[3]
    hb = C.sizeof(SliceHeader)
    sh = SliceHeader.from_buffer_copy(bytes(ref.headers)[hb:2 * hb])
    r = ref.results[1]
    ctx = Ctx(sa, pa)
    ctx.slice(1, 1, SliceHeader())
    bs, buf = new_bs()
    assert H.h264_write_nalu(C.byref(bs), ctx.p) == 0
    assert H.h264_ctx_clear(ctx.p) == 0
    assert H.h264_ctx_set_sps(ctx.p, C.byref(sb)) == 0
    assert H.h264_ctx_set_pps(ctx.p, C.byref(pb)) == 0
    ctx.slice(r.nal_ref_idc, r.nal_unit_type, sh)
    bs2, buf2 = new_bs()
    assert H.h264_write_nalu(C.byref(bs2), ctx.p) == 0
    assert 8 * bs2.off + bs2.cachebits == r.hdr_len_bits
    assert buf2.raw[:bs2.off] == nal[:bs2.off]
    ctx.close()


@pytest.mark.parametrize("cabac", [0, 1])
def test_single_nal_header_on_the_host_equals_device_path(monkeypatch, cabac):
    """h264_reader_parse_nalu(flags = 0) reads a slice header on the host
    (csrc/host/slicehdr.c); with H264_AMD_NO_HOST_HEADERS the same call goes
    through the kernel.  Both must deliver the oracle's event log, statuses and
    header images -- clean and damaged NAL units, header extras (reordering,
    weights, marking), field pictures, and CABAC pictures with SLICE_DATA set
    (no slice data walk there, src/h264_syntax_slice_data.h:715-717)."""
    kw = dict(weighted_pred_flag=1, weighted_bipred_idc=1, field_coding=1,
              pic_order_cnt_type=1,
              bottom_field_pic_order_in_frame_present_flag=1)
    if cabac:
        kw.update(entropy_coding_mode_flag=1, profile_idc=100)
    cfg = streams.small_cfg(**kw)
    nals = mixed_nals(cfg, n_slices=30, seed0=200)
    if not cabac:
        nals = streams.damage(nals, 11)
    nals.insert(7, b"\x65")
    nals.insert(9, b"\x41\x9a")
    flags = 1 if cabac else 0
    ref = orc.parse_nalus(nals, flags=flags, want=0)
    host = prod.parse_nalus(nals, flags=flags)
    monkeypatch.setenv("H264_AMD_NO_HOST_HEADERS", "1")
    dev = prod.parse_nalus(nals, flags=flags)
    for got in (host, dev):
        assert got.statuses == ref.statuses
        same_log(ref, got, stream=False)
    assert any(s < 0 for s in ref.statuses) or cabac
    assert sum(1 for e in ref.event_tuples() if e[0] == 4) >= 20


@pytest.mark.parametrize("case", [0, 2, 3, 6])
def test_reader_slice_groups(case):
    """FMO through the drop-in reader: slice_data_mb callbacks carry the
    addresses of the slice's group (h264_next_mb_addr,
    ref src/h264_fmo.c:307-319), the host walks the map the kernel built."""
    from test_oracle_vs_generator import fmo_cfg, fmo_slices
    cfg, cycle = fmo_cfg(case)
    nals = [gen.sps(cfg), gen.pps(cfg)]
    want_addrs = []
    for s, addrs in fmo_slices(cfg, cycle, 950):
        nals.append(gen.slice_nal(cfg, s))
        want_addrs += addrs.tolist()
    data = streams.annexb(nals)
    ref = orc.parse_stream(data, want=0)
    got = prod.parse_stream(data)
    same_log(ref, got, stream=True)
    assert [e[1] for e in got.event_tuples() if e[0] == 7] == want_addrs


def test_reader_mbaff():
    """MBAFF through the drop-in reader: addresses of the callbacks count
    macroblocks, two per pair, from 2 * first_mb_in_slice
    (ref src/h264_syntax_slice_data.h:725); with slice groups on top the map
    units are pairs."""
    from test_oracle_vs_generator import fmo_cfg, mbaff_slices
    cfg = streams.small_cfg(field_coding=2)
    nals = [gen.sps(cfg), gen.pps(cfg)]
    nals += [gen.slice_nal(cfg, s)
             for s in mbaff_slices(cfg, 1700, n=8, pm_skip=300)]
    data = streams.annexb(nals)
    ref = orc.parse_stream(data, want=0)
    got = prod.parse_stream(data)
    same_log(ref, got, stream=True)
    assert any(e[0] == 7 for e in got.event_tuples())

    cfg, cycle = fmo_cfg(1, field_coding=2)
    m = gen.slice_group_map(cfg, cycle, 0)
    nals = [gen.sps(cfg), gen.pps(cfg)]
    want = []
    for g in sorted(set(m.tolist())):
        addrs = np.nonzero(m == g)[0]
        nals.append(gen.slice_nal(cfg, streams.rich_slice(
            cfg, 1800 + g, (2, 0, 1)[g % 3], int(addrs[0]) // 2, len(addrs),
            field_pic=0)))
        want += addrs.tolist()
    data = streams.annexb(nals)
    ref = orc.parse_stream(data, want=0)
    got = prod.parse_stream(data)
    same_log(ref, got, stream=True)
    assert [e[1] for e in got.event_tuples() if e[0] == 7] == want
