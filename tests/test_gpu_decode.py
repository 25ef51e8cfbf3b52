"""GPU parity: slice decode kernel vs the oracle (and vs the generator's own
expectation) through the C-ABI.  Bit-exact: integer/byte work."""
import ctypes as C

import numpy as np
import pytest

import orc
import streams
from libh264_amd import gen, Params
from libh264_amd.ctypes_abi import (Sps, Pps, SliceHeader, SliceResult,
                                    TIER_NONE, TIER_T0, TIER_T1, TIER_T2,
                                    t2_stride)

pytestmark = pytest.mark.gpu


def result_fields(r):
    return {n: getattr(r, n) for n, _ in SliceResult._fields_}


def make_params(oracle_rec):
    p = Params()
    for ev in oracle_rec.events:
        if ev.kind == 8:
            p.put_sps(oracle_rec.sps[ev.a])
        elif ev.kind == 9:
            p.put_pps(oracle_rec.pps[ev.a])
    return p


def run_case(engine, cfg, slices, tier, lead=0, flags=1):
    sps, pps = gen.sps(cfg), gen.pps(cfg)
    nals, expect = [], []
    for s in slices:
        n, e = gen.slice_nal(cfg, s, want_t2=True)
        nals.append(n)
        expect.append(e)
    ref = orc.parse_nalus([sps, pps] + nals, flags=flags)
    assert ref.statuses[0] == 0 and ref.statuses[1] == 0
    params = make_params(ref)
    buf, offs, sizes = streams.pack_nals(nals, lead=lead)
    d_buf = engine.upload(buf)
    descs = [(o, s, params.epoch) for o, s in zip(offs, sizes)]
    cap = max(s.num_mbs for s in slices) + 2
    out = engine.decode_slices(params, d_buf, descs, reader_flags=flags,
                               tier=tier, mb_capacity=cap, want_headers=True)
    res = engine.fetch_results(out)
    hdrs = engine.fetch_headers(out)
    for i, s in enumerate(slices):
        want = result_fields(ref.results[i])
        got = result_fields(res[i])
        assert got == want, (i, got, want)
        assert bytes(hdrs[i]) == bytes(ref.headers[i]), i
        if flags == 0:
            continue
        assert res[i].mb_count == s.num_mbs
        if tier == TIER_T2:
            mb = engine.fetch_mb(out, i, res[i].mb_count)
            st = t2_stride(res[i].chroma_array_type)
            got_mb = mb[:, :st]
            assert np.array_equal(got_mb, ref.mb_records(i)), i
            assert np.array_equal(got_mb, expect[i]), i
        elif tier == TIER_T1:
            mb = engine.fetch_mb(out, i, res[i].mb_count)
            assert np.array_equal(mb, ref.mb_records(i)[:, :64]), i
        elif tier == TIER_T0:
            mb = engine.fetch_mb(out, i, res[i].mb_count)
            assert np.array_equal(mb, ref.mb_records(i)[:, :4]), i
    return res


@pytest.mark.parametrize("tier", [TIER_T2, TIER_T1, TIER_T0, TIER_NONE])
def test_ipb_slices_420(engine, tier):
    cfg = streams.small_cfg()
    slices = [streams.rich_slice(cfg, seed, st,
                                 first_mb=0 if seed % 3 else 5)
              for seed in range(24) for st in (2, 0, 1)]
    run_case(engine, cfg, slices, tier)


@pytest.mark.parametrize("cf,bd", [(0, 0), (2, 0), (3, 0), (1, 2), (3, 4),
                                   (1, 6), (3, 6), (0, 5)])
def test_chroma_formats(engine, cf, bd):
    cfg = streams.small_cfg(cf=cf, bd=bd, weighted_pred_flag=1,
                            weighted_bipred_idc=1,
                            num_ref_idx_l1_default_minus1=2,
                            direct_8x8_inference_flag=0,
                            deblocking_filter_control_present_flag=0)
    slices = [streams.rich_slice(cfg, seed, st + 5 * (seed % 2),
                                 first_mb=seed % 7, pm_pcm=60)
              for seed in range(12) for st in (2, 0, 1)]
    run_case(engine, cfg, slices, TIER_T2)


@pytest.mark.parametrize("lead", [1, 2, 3])
def test_unaligned_nal_start(engine, lead):
    cfg = streams.small_cfg()
    slices = [streams.rich_slice(cfg, 100 + seed, 2) for seed in range(6)]
    run_case(engine, cfg, slices, TIER_T2, lead=lead)


def test_header_only(engine):
    cfg = streams.small_cfg()
    slices = [streams.rich_slice(cfg, seed, st)
              for seed in range(8) for st in (2, 0, 1)]
    run_case(engine, cfg, slices, TIER_NONE, flags=0)


def test_walk_without_macroblock_output_needs_no_capacity(engine):
    """reader_flags = SLICE_DATA, tier NONE and mb_capacity left at 0 walks
    the slice data (include/h264_amd.h: 0 = no limit without macroblock
    output); it used to end every slice with H264_AMD_E_MB_OVERFLOW."""
    cfg = streams.small_cfg()
    slices = [streams.rich_slice(cfg, seed, st)
              for seed in range(6) for st in (2, 0, 1)]
    nals = [gen.slice_nal(cfg, s) for s in slices]
    ref = orc.parse_nalus([gen.sps(cfg), gen.pps(cfg)] + nals, flags=1)
    params = make_params(ref)
    buf, offs, sizes = streams.pack_nals(nals)
    d_buf = engine.upload(buf)
    descs = [(o, s, params.epoch) for o, s in zip(offs, sizes)]
    out = engine.decode_slices(params, d_buf, descs, reader_flags=1,
                               tier=TIER_NONE)
    res = engine.fetch_results(out)
    for i, s in enumerate(slices):
        assert result_fields(res[i]) == result_fields(ref.results[i]), i
        assert res[i].status == 0 and res[i].mb_count == s.num_mbs


def test_1080p_slice(engine):
    cfg = gen.cfg_1080p()
    slices = [streams.rich_slice(cfg, 7, 7, pm_pcm=2, density=5,
                                 pm_escape=5, hdr_extras=0)]
    run_case(engine, cfg, slices, TIER_T2)


def test_api_misuse_is_refused_on_the_host(engine):
    """Bad pointers / strides must come back as -EINVAL from the entry point,
    never reach the kernel (a faulting kernel can take the whole host's GPUs
    down)."""
    import ctypes as C
    from libh264_amd import capi
    from libh264_amd.ctypes_abi import DecodeArgs
    L = capi.engine_lib()
    cfg = streams.small_cfg()
    nal = gen.slice_nal(cfg, streams.rich_slice(cfg, 1, 2))
    ref = orc.parse_nalus([gen.sps(cfg), gen.pps(cfg), nal])
    params = make_params(ref)
    buf, offs, sizes = streams.pack_nals([nal])
    d_buf = engine.upload(buf)
    good = engine.decode_slices(params, d_buf, [(offs[0], sizes[0],
                                                 params.epoch)],
                                reader_flags=1, tier=TIER_T2, mb_capacity=64,
                                want_headers=True)

    def args(**kw):
        a = DecodeArgs()
        a.d_buf = d_buf.ptr
        a.d_descs = good["d_descs"].ptr
        a.n = 1
        a.reader_flags = 1
        a.tier = TIER_T2
        a.mb_capacity = 64
        a.d_results = good["results"].ptr
        a.d_headers = good["headers"].ptr
        a.d_mb_out = good["mb"].ptr
        a.mb_stride = t2_stride(3)
        for k, v in kw.items():
            setattr(a, k, v)
        return a

    def call(a):
        return L.h264_amd_decode_slices(engine.h, params.h, C.byref(a))

    assert call(args()) == 0
    assert call(args(d_buf=None)) == -22
    assert call(args(d_results=good["results"].ptr.value + 4)) == -22
    assert call(args(d_headers=good["headers"].ptr.value + 8)) == -22
    assert call(args(d_mb_out=good["mb"].ptr.value + 4)) == -22
    assert call(args(mb_stride=t2_stride(3) - 16)) == -22
    assert call(args(mb_stride=t2_stride(3) + 4)) == -22
    assert call(args(tier=7)) == -22
    assert call(args(mb_capacity=0)) == -22
    assert call(args(n=0)) == 0
    assert L.h264_amd_decode_slices(None, params.h, C.byref(args())) == -22
    assert L.h264_amd_decode_slices(engine.h, None, C.byref(args())) == -22
    engine.sync()
    res = engine.fetch_results(good)
    assert res[0].status == 0


def test_very_wide_picture(engine):
    """300 macroblocks per row: the row context pushes the workgroup's LDS past
    48 KiB (dynamic shared memory attribute path)."""
    cfg = streams.small_cfg(width_mbs=300, height_mbs=3)
    slices = [streams.rich_slice(cfg, 5, 2), streams.rich_slice(cfg, 6, 0),
              streams.rich_slice(cfg, 7, 1, 299, 400)]
    run_case(engine, cfg, slices, TIER_T2)


@pytest.mark.parametrize("fn4,poc4", [
    (28, 2),            # 32-bit frame_num
    (29, 28),           # 33 / 32 bits: the value is preceded by a zero
    (60, 29),           # 64 / 33 bits
    (61, 2),            # 65 bits: parses, cannot be written back (-EINVAL)
    (2, 61),
    (300, 2),           # low byte of the width would be 48
    (130560, 2),        # low byte 4: longer than most of the slices
    (39996, 2),         # 5000 bytes: lands in the macroblock data of some
    (0xfffffffc, 2),    # 4 + minus4 wraps to 0: no bits at all
    (0xfffffffd, 0xfffffffc),
])
def test_hostile_sps_field_widths(engine, fn4, poc4):
    """The reference checks neither log2_max_frame_num_minus4 nor
    log2_max_pic_order_cnt_lsb_minus4 (src/h264_syntax.h:309-313): its reader
    takes any u(n) (all n bits consumed, last 32 kept), its writer up to 64 bits.
    Read and write kernels must do the same, not truncate the width."""
    cfg = streams.small_cfg(log2_max_frame_num_minus4=fn4,
                            log2_max_poc_lsb_minus4=poc4)
    slices = [streams.rich_slice(cfg, 40 + seed, st, frame_num=0x9abcdef1,
                                 poc_lsb=0xfedcba98, pm_pcm=pcm)
              for seed in range(4) for st in (2, 0, 1) for pcm in (0, 900)]
    nals = [gen.slice_nal(cfg, s) for s in slices]
    ref = orc.parse_nalus([gen.sps(cfg), gen.pps(cfg)] + nals, flags=1)
    # separately: a failed write-back zeroes the context's header length
    # (ref src/h264_syntax.h:1250), which the result record would then show
    ref_w = orc.parse_nalus([gen.sps(cfg), gen.pps(cfg)] + nals, flags=1,
                            want=orc.ORC_REC_WANT_REWRITE)
    assert ref.statuses[0] == 0 and ref.statuses[1] == 0
    assert ref.sps[0].log2_max_frame_num_minus4 == fn4
    if 4 + fn4 <= 40000 or 4 + fn4 >= 1 << 32:
        assert ref.headers[0].frame_num == (0x9abcdef1 & ((1 << min(
            32, (fn4 + 4) & 0xffffffff)) - 1))
    params = make_params(ref)
    buf, offs, sizes = streams.pack_nals(nals)
    d_buf = engine.upload(buf)
    descs = [(o, s, params.epoch) for o, s in zip(offs, sizes)]
    out = engine.decode_slices(params, d_buf, descs, reader_flags=1,
                               tier=TIER_T2, mb_capacity=64, want_headers=True)
    res = engine.fetch_results(out)
    hdrs = engine.fetch_headers(out)
    good = 0  # the oracle keeps a header per slice whose callback fired
    for i in range(len(slices)):
        assert res[i].status == ref.statuses[i + 2], i
        assert result_fields(res[i]) == result_fields(ref.results[i]), i
        if res[i].status == 0:
            assert bytes(hdrs[i]) == bytes(ref.headers[good]), i
            good += 1
            mb = engine.fetch_mb(out, i, res[i].mb_count)
            st = t2_stride(res[i].chroma_array_type)
            assert np.array_equal(mb[:, :st], ref.mb_records(i)), i
    if all((4 + x) & 0xffffffff <= 40000 for x in (fn4, poc4)):
        assert good == len(slices)
    else:
        assert good < len(slices)
    # write-back: same bytes, or the same refusal
    d_out, ooffs, osizes = engine.write_slices(params, d_buf, out,
                                               max(sizes) + 64)
    raw = engine.download(d_out)
    rew = {w.nal_index: w for w in ref_w.rewrite_idx}
    for i in range(len(slices)):
        if res[i].status != 0:
            continue
        w = rew[i + 2]
        if w.res < 0:
            assert osizes[i] == w.res, (i, osizes[i], w.res)
            continue
        assert osizes[i] == w.len, (i, osizes[i], w.len)
        got = raw[int(ooffs[i]):int(ooffs[i]) + w.len].tobytes()
        assert got == ref_w.rewrites[w.off:w.off + w.len], i


def test_absurd_picture_widths(engine):
    """A damaged SPS can carry any 32-bit width (the drop-in fuzz found one of
    65408 pixels that made the whole batch unlaunchable, -E2BIG).  Slice data
    is walked up to H264_AMD_MAX_PIC_WIDTH_MBS columns; wider pictures and
    width 0 (only reachable through h264_ctx_set_sps; the reference divides
    by it) end with
    H264_AMD_E_UNSUPPORTED, their headers parsed, other slices of the batch
    untouched -- and a header-only parse does not care about the width."""
    from libh264_amd.ctypes_abi import SliceHeader
    b = orc.Builder()
    cfg = streams.small_cfg()
    good = [gen.slice_nal(cfg, streams.rich_slice(cfg, s, 2)) for s in range(3)]
    nals = [gen.sps(cfg), gen.pps(cfg)] + good
    for sps_id, wm1 in ((1, 1055), (2, 1056), (3, 4087), (4, 0xfffffffe)):
        sps = orc.parse_nalus([gen.sps(cfg)]).sps[0]
        sps = type(sps).from_buffer_copy(bytes(sps))
        sps.seq_parameter_set_id = sps_id
        sps.pic_width_in_mbs_minus1 = wm1
        sps.pic_height_in_map_units_minus1 = 0
        pps = orc.parse_nalus([gen.sps(cfg), gen.pps(cfg)]).pps[0]
        pps = type(pps).from_buffer_copy(bytes(pps))
        pps.pic_parameter_set_id = 10 + sps_id
        pps.seq_parameter_set_id = sps_id
        nals += [b.nalu(3, 7, sps=sps), b.nalu(3, 8, sps=sps, pps=pps)]
        # a slice of that picture: take a good slice's bytes, re-head it
        donor = streams.rich_slice(cfg, 9, 2, num_mbs=6, hdr_extras=0)
        c2 = streams.small_cfg(width_mbs=8, height_mbs=6)
        c2.pps_id = 10 + sps_id
        nals.append(gen.slice_nal(c2, donor))
    for flags in (1, 0):
        ref = orc.parse_nalus(nals, flags=flags)
        assert all(s == 0 for s in ref.statuses[:5])
        sl = [n for n in nals if (n[0] & 0x1f) in (1, 5)]
        params = make_params(ref)
        buf, offs, sizes = streams.pack_nals(sl)
        d_buf = engine.upload(buf)
        descs = [(o, s, params.epoch) for o, s in zip(offs, sizes)]
        out = engine.decode_slices(params, d_buf, descs, reader_flags=flags,
                                   tier=TIER_T2 if flags else TIER_NONE,
                                   mb_capacity=64, want_headers=True)
        res = engine.fetch_results(out)
        got = [result_fields(r) for r in res]
        want = [result_fields(r) for r in ref.results]
        assert got == want
        st = [r.status for r in res]
        if flags:
            assert st[:3] == [0, 0, 0] and st[4:] == [-38, -38, -38], st
            for i in range(3):
                mb = engine.fetch_mb(out, i, res[i].mb_count)
                assert np.array_equal(mb[:, :t2_stride(1)], ref.mb_records(i))
        else:
            assert st == [0] * 7


@pytest.mark.parametrize("tier", [TIER_T2, TIER_T0])
def test_field_pictures(engine, tier):
    """Interlaced material without MBAFF: field pictures (field_pic_flag = 1,
    top and bottom) carry the field flag on every macroblock record, skipped
    ones included (ref src/h264_slice_data.c:1145-1149); frame pictures of the
    same stream do not; with mb_adaptive_frame_field_flag frame pictures are
    MBAFF (test_mbaff_*) and field pictures decode as before.  Found by
    tools/fuzz_params.py: the kernels used to leave the flag clear."""
    cfg = streams.small_cfg(field_coding=1,
                            bottom_field_pic_order_in_frame_present_flag=1)
    slices = [streams.rich_slice(cfg, 60 + seed, st, field_pic=fp)
              for seed in range(6) for st in (2, 0, 1) for fp in (0, 1, 2)]
    res = run_case(engine, cfg, slices, tier)
    __import__("test_gpu_scan_write").roundtrip_case(engine, cfg, slices)
    cfg = streams.small_cfg(field_coding=2)
    slices = [streams.rich_slice(cfg, 80 + seed, st, field_pic=fp)
              for seed in range(3) for st in (2, 0, 1) for fp in (1, 2)]
    run_case(engine, cfg, slices, tier)


@pytest.mark.parametrize("cf", [1, 3])
def test_sp_and_si_slices(engine, cf):
    cfg = streams.small_cfg(cf=cf, weighted_pred_flag=1)
    slices = [streams.rich_slice(cfg, 70 + seed, st, first_mb=seed % 5)
              for seed in range(8) for st in (3, 4, 8, 9)]
    run_case(engine, cfg, slices, TIER_T2)
    roundtrip = __import__("test_gpu_scan_write").roundtrip_case
    roundtrip(engine, cfg, slices[:12])


@pytest.mark.parametrize("always_zero,field_coding", [(0, 0), (0, 1), (1, 1)])
def test_pic_order_cnt_type_1(engine, always_zero, field_coding):
    cfg = streams.small_cfg(pic_order_cnt_type=1,
                            delta_pic_order_always_zero_flag=always_zero,
                            bottom_field_pic_order_in_frame_present_flag=1,
                            field_coding=field_coding)
    slices = [streams.rich_slice(cfg, 70 + seed, st,
                                 field_pic=seed % 3 if field_coding else 0)
              for seed in range(8) for st in (2, 0, 1)]
    run_case(engine, cfg, slices, TIER_T2)
    __import__("test_gpu_scan_write").roundtrip_case(engine, cfg, slices)


def test_pps_id_read_across_the_end_of_the_nal(engine):
    """pic_parameter_set_id whose Exp-Golomb suffix is cut off by the end of
    the NAL unit: -EIO from the read (as in the reference), not -EINVAL from
    the value the zero padding produces.  Found by tools/fuzz_dropin.py."""
    cfg = streams.small_cfg()
    nals = [bytes.fromhex("210586000000da25"), bytes.fromhex("2105860000"),
            bytes.fromhex("21058600000001"), bytes.fromhex("2588"),
            bytes.fromhex("25")]
    ref = orc.parse_nalus([gen.sps(cfg), gen.pps(cfg)] + nals)
    assert ref.statuses[2] == -5
    params = make_params(ref)
    buf, offs, sizes = streams.pack_nals(nals)
    d_buf = engine.upload(buf)
    descs = [(o, s, params.epoch) for o, s in zip(offs, sizes)]
    out = engine.decode_slices(params, d_buf, descs, reader_flags=1,
                               tier=TIER_T2, mb_capacity=64, want_headers=True)
    res = engine.fetch_results(out)
    for i in range(len(nals)):
        assert result_fields(res[i]) == result_fields(ref.results[i]), i


def test_descriptors_outside_the_buffer_are_answered_not_followed(engine):
    """with d_buf_len / d_src_len given, a descriptor that reaches past the
    buffer costs its slice H264_AMD_E_BAD_DESC (-EFAULT); the kernels never
    dereference it (a GPU memory fault can take the whole host's GPUs down)"""
    cfg = streams.small_cfg()
    nals = [gen.slice_nal(cfg, streams.rich_slice(cfg, s, 2)) for s in range(3)]
    ref = orc.parse_nalus([gen.sps(cfg), gen.pps(cfg)] + nals)
    params = make_params(ref)
    buf, offs, sizes = streams.pack_nals(nals)
    d_buf = engine.upload(buf)
    n = d_buf.nbytes
    descs = [(offs[0], sizes[0], params.epoch),
             (n + 4096, 100, params.epoch),           # starts behind the end
             (offs[1], n, params.epoch),              # runs past the end
             ((1 << 64) - 8, 64, params.epoch),       # offset + size wraps
             (n - 4, 8, params.epoch),                # straddles the end
             (offs[2], sizes[2], params.epoch)]
    out = engine.decode_slices(params, d_buf, descs, reader_flags=1,
                               tier=TIER_T2, mb_capacity=64, want_headers=True)
    res = engine.fetch_results(out)
    assert [r.status for r in res] == [0, -14, -14, -14, -14, 0]
    assert [r.mb_count for r in res] == [48, 0, 0, 0, 0, 48]
    for i, k in ((0, 0), (5, 2)):
        assert result_fields(res[i]) == result_fields(ref.results[k])
    d_out, ooffs, osizes = engine.write_slices(params, d_buf, out,
                                               max(sizes) + 64)
    assert int(osizes[0]) == sizes[0] and int(osizes[5]) == sizes[2]
    assert all(int(osizes[i]) < 0 for i in (1, 2, 3, 4))


@pytest.mark.parametrize("case", [0, 1, 2, 3, "3r", 4, 5, 6])
def test_slice_groups(engine, case):
    """FMO, every slice group map type: the kernel builds the map behind the
    slice header, walks the addresses of the slice's group and takes
    neighbour availability from them; full records vs oracle and vs the
    generator's own expectation, and the map itself vs the generator's."""
    from test_oracle_vs_generator import fmo_cfg, fmo_slices
    cfg, cycle = fmo_cfg(case)
    pairs = fmo_slices(cfg, cycle, 700)
    slices = [s for s, _ in pairs]
    run_case(engine, cfg, slices, TIER_T2)
    run_case(engine, cfg, slices, TIER_T0)
    # the map the kernel built for the last slice of the launch
    from libh264_amd import capi
    L = capi.engine_lib()
    L.h264_amd_slice_group_map.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p,
                                           C.c_size_t]
    n = cfg.width_mbs * cfg.height_mbs
    got = np.zeros(n, dtype=np.uint8)
    assert L.h264_amd_slice_group_map(engine.h, len(slices) - 1,
                                      got.ctypes.data_as(C.c_void_p), n) == 0
    assert np.array_equal(got, gen.slice_group_map(cfg, cycle))
    assert L.h264_amd_slice_group_map(engine.h, len(slices),
                                      got.ctypes.data_as(C.c_void_p),
                                      n) == -34  # -ERANGE


@pytest.mark.parametrize("field_pic", [0, 1, 2])
def test_slice_groups_interlaced(engine, field_pic):
    from test_oracle_vs_generator import fmo_cfg, fmo_slices
    cfg, cycle = fmo_cfg(1, field_coding=1)
    pairs = fmo_slices(cfg, cycle, 800, field_pic=field_pic)
    run_case(engine, cfg, [s for s, _ in pairs], TIER_T2)


def test_slice_groups_box_out_1080p(engine):
    """box-out over a 1080p picture: 8160 map units walked on the LDS image,
    two slices that together code the picture"""
    from test_oracle_vs_generator import FMO_CASES  # noqa: F401
    cfg = gen.cfg_1080p()
    cfg.num_slice_groups_minus1 = 1
    cfg.slice_group_map_type = 3
    cfg.slice_group_change_direction_flag = 1
    cfg.slice_group_change_rate_minus1 = 99
    cycle = 37
    slices = []
    for k, (first, n, _a) in enumerate(gen.group_slices(cfg, cycle)):
        slices.append(streams.rich_slice(cfg, 900 + k, 7, first, n,
                                         slice_group_change_cycle=cycle,
                                         density=2, hdr_extras=0))
    assert sum(s.num_mbs for s in slices) == 8160
    run_case(engine, cfg, slices, TIER_T0)


@pytest.mark.parametrize("cf", [1, 2, 3, 0])
def test_mbaff_frames(engine, cf):
    """MBAFF on the device: macroblock pairs, the field decoding flag read or
    inferred, neighbours per 6.4.12.2 (ref src/h264_macroblock.c:110-231),
    doubled ref_idx range in field macroblocks; every record vs the oracle
    and vs the generator's expectation."""
    from test_oracle_vs_generator import mbaff_slices
    cfg = streams.small_cfg(cf=cf, field_coding=2)
    slices = mbaff_slices(cfg, 1100)
    run_case(engine, cfg, slices, TIER_T2)
    run_case(engine, cfg, slices, TIER_T0)


def test_mbaff_skip_heavy_and_mixed_with_field_pictures(engine):
    from test_oracle_vs_generator import mbaff_slices
    cfg = streams.small_cfg(field_coding=2,
                            bottom_field_pic_order_in_frame_present_flag=1)
    slices = mbaff_slices(cfg, 1200, n=12, pm_skip=600, end_with_skip=1)
    slices += mbaff_slices(cfg, 1300, n=6, pm_skip=850)
    slices += [streams.rich_slice(cfg, 1400 + k, (2, 0, 1)[k % 3],
                                  field_pic=1 + k % 2) for k in range(4)]
    run_case(engine, cfg, slices, TIER_T2)
    run_case(engine, cfg, slices, TIER_T1)


def test_mbaff_with_slice_groups(engine):
    from test_oracle_vs_generator import fmo_cfg
    cfg, cycle = fmo_cfg(1, field_coding=2)
    m = gen.slice_group_map(cfg, cycle, 0)
    slices = []
    for g in sorted(set(m.tolist())):
        addrs = np.nonzero(m == g)[0]
        slices.append(streams.rich_slice(cfg, 1500 + g, (2, 0, 1)[g % 3],
                                         int(addrs[0]) // 2, len(addrs),
                                         field_pic=0))
    run_case(engine, cfg, slices, TIER_T2)


def test_mbaff_1080i(engine):
    """a 1080-line MBAFF frame (120 x 34 pairs): the 32-byte rows of the pair
    walk at full width, I and P"""
    cfg = gen.cfg_1080p(height_mbs=34, field_coding=2, profile_idc=77,
                        crop_bottom=2)
    slices = [streams.rich_slice(cfg, 1600 + k, (2, 0)[k], 0,
                                 cfg.width_mbs * cfg.height_mbs * 2,
                                 field_pic=0, density=2, hdr_extras=0)
              for k in range(2)]
    run_case(engine, cfg, slices, TIER_T0)
    run_case(engine, cfg, slices, TIER_T2)


def test_slice_group_limits(engine):
    """slice groups on pictures beyond what the map scratch is built for
    (include/h264_amd.h, H264_AMD_FMO_MAX_PIC_MBS / _BOX_OUT_UNITS): slice data
    refused with -ENOSYS by kernel and oracle alike, header still parsed; one
    macroblock row below the limit decodes; type 0 with the longest run length
    the PPS syntax can carry (the reference's 32-bit sum wraps there)."""
    from test_oracle_vs_generator import fmo_cfg

    def run(cfg, cycle=0, n=10):
        s = streams.rich_slice(cfg, 2100, 2, 0, n,
                               slice_group_change_cycle=cycle, hdr_extras=0)
        nals = [gen.sps(cfg), gen.pps(cfg), gen.slice_nal(cfg, s)]
        ref = orc.parse_nalus(nals, flags=1)
        params = make_params(ref)
        buf, offs, sizes = streams.pack_nals(nals[2:])
        out = engine.decode_slices(
            params, engine.upload(buf), [(offs[0], sizes[0], params.epoch)],
            reader_flags=1, tier=TIER_T0, mb_capacity=n + 2,
            want_headers=True)
        res = engine.fetch_results(out)
        assert result_fields(res[0]) == result_fields(ref.results[0])
        return res

    # 1056 x 132 = 139392 macroblocks > 139264; 1056 x 131 fits
    cfg, _ = fmo_cfg(1)
    cfg.width_mbs, cfg.height_mbs = 1056, 132
    res = run(cfg)
    assert res[0].status == -38 and res[0].flags & 1  # header parsed
    cfg.height_mbs = 131
    res = run(cfg)
    assert res[0].status == 0 and res[0].mb_count == 10
    # box-out: 182 x 181 = 32942 map units > 32768; 181 x 181 = 32761 fits
    cfg, cycle = fmo_cfg(3)
    cfg.width_mbs, cfg.height_mbs = 182, 181
    assert run(cfg, cycle)[0].status == -38
    cfg.width_mbs = 181
    assert run(cfg, cycle)[0].status == 0
    # interleaved with run_length_minus1 = 2^32 - 2
    cfg, _ = fmo_cfg(0)
    cfg.run_length_minus1[1] = 0xfffffffe
    assert run(cfg)[0].status == 0


def test_box_out_of_an_extreme_aspect_ratio(engine):
    """1056 x 6 macroblocks: once the picture's edge clips the box, a lap of the
    box-out walk costs two thousand steps and fills a dozen units; kernel and
    oracle stop after H264_AMD_FMO_BOX_OUT_STEPS steps and agree on the map
    that leaves (found by tools/fuzz_dropin.py params, seed 61733: the oracle
    had no bound).  The addresses of the slice's macroblocks come out of the
    map: compared through the drop-in's callbacks in test_gpu_dropin.py; here
    the records and the map's effect on the neighbour availability."""
    from test_oracle_vs_generator import fmo_cfg
    cfg, _ = fmo_cfg(3)
    cfg.width_mbs, cfg.height_mbs = 1056, 6
    cfg.slice_group_change_rate_minus1 = 0
    for cycle, first in ((900, 0), (2500, 0), (6336, 0), (900, 2000)):
        s = streams.rich_slice(cfg, 2300 + cycle, 2, first, 400,
                               slice_group_change_cycle=cycle, hdr_extras=0)
        nals = [gen.sps(cfg), gen.pps(cfg), gen.slice_nal(cfg, s)]
        ref = orc.parse_nalus(nals, flags=1)
        params = make_params(ref)
        buf, offs, sizes = streams.pack_nals(nals[2:])
        out = engine.decode_slices(
            params, engine.upload(buf), [(offs[0], sizes[0], params.epoch)],
            reader_flags=1, tier=TIER_T2, mb_capacity=402, want_headers=True)
        res = engine.fetch_results(out)
        assert result_fields(res[0]) == result_fields(ref.results[0]), cycle
