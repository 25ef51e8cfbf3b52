"""Pins of the ORACLE (oracle/, the CPU restatement used as checker).

The reference has no fixtures or golden vectors of its own and
cannot be compiled here without stand-ins for <ulog.h>, so the oracle is pinned
by what IS available:
  * the known answers the survey recorded from a run of the real reference
    (BASELINE.md section 2): parameter-set bytes, the
    sizes of the synthetic slices, the dense-CAVLC recipe and its size;
  * the reference's own inline bit reader, the one piece that compiles from
    /root/reference/include alone (oracle/_ref/libref_bits.so);
  * struct sizes measured on the reference's headers;
  * an independent second implementation: the CAVLC *encoder* of
    libh264_amd/csrc/gen, whose expected records must equal what the oracle
    parses back (tests/test_oracle_vs_generator.py).
"""
import ctypes as C
import hashlib
import os
import random

import numpy as np
import pytest

import orc
import streams
from h264types import (Sps, Pps, SliceHeader, REFERENCE_SIZEOF)
from libh264_amd import gen

H264_MB_TYPE_I_NxN = 1  # enum h264_mb_type, include/h264/h264_types.h


def ps_1080p(entropy, profile=100):
    sps = Sps()
    sps.profile_idc = profile
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


def test_struct_sizes_match_reference_headers():
    L = orc.lib()
    for name, size in REFERENCE_SIZEOF.items():
        assert L.orc_sizeof(name.encode()) == size, name
    assert C.sizeof(Sps) == REFERENCE_SIZEOF["h264_sps"]
    assert C.sizeof(Pps) == REFERENCE_SIZEOF["h264_pps"]
    assert C.sizeof(SliceHeader) == REFERENCE_SIZEOF["h264_slice_header"]


def test_parameter_set_known_bytes():
    """bytes the reference's h264_write_nalu produced."""
    sps, pps = ps_1080p(1)
    b = orc.Builder()
    assert b.nalu(3, 7, sps=sps).hex() == "67640028acb403c01132"
    assert b.nalu(3, 8, sps=sps, pps=pps).hex() == "68ee3c80"
    # and back
    back = Sps()
    raw = bytes.fromhex("67640028acb403c01132")
    assert orc.lib().orc_parse_sps(raw, len(raw), C.byref(back)) == 0
    assert bytes(back) == bytes(sps)


def test_synthetic_slice_known_sizes():
    """BASELINE.md section 2 / src/h264_writer.c:79-237, measured on the
    reference at 8160 macroblocks."""
    b = orc.Builder()
    sh = SliceHeader()
    sh.slice_type = 7
    sps, pps = ps_1080p(0)
    assert len(b.nalu(3, 5, sps=sps, pps=pps, sh=sh, kind=1,
                      mb_count=8160)) == 8164
    sps, pps = ps_1080p(1)
    assert len(b.nalu(3, 5, sps=sps, pps=pps, sh=sh, kind=1,
                      mb_count=8160)) == 402
    sh = SliceHeader()
    sh.slice_type = 5
    assert len(b.nalu(2, 1, sps=sps, pps=pps, sh=sh, kind=2,
                      mb_count=8160)) == 63


def test_survey_probe_stream_size():
    """the survey's probe stream -- SPS, PPS and 64 grey
    1080p IDR slices behind four-byte start codes -- was 522 774 bytes on the
    real reference; the oracle's writer builds the same number of bytes and its
    reader walks 64 x 8160 macroblocks out of them."""
    b = orc.Builder()
    sps, pps = ps_1080p(0)
    sh = SliceHeader()
    sh.slice_type = 7
    nals = [b.nalu(3, 7, sps=sps), b.nalu(3, 8, sps=sps, pps=pps)]
    for k in range(64):
        sh.idr_pic_id = 0
        nals.append(b.nalu(3, 5, sps=sps, pps=pps, sh=sh, kind=1,
                           mb_count=8160))
    data = b"".join(b"\x00\x00\x00\x01" + n for n in nals)
    assert len(data) == 522774
    rec = orc.parse_stream(data, want=0)
    assert rec.consumed == len(data) and len(rec.nals) == 66
    assert [r.mb_count for r in rec.results] == [8160] * 64
    assert all(r.status == 0 for r in rec.results)


class Bits:
    def __init__(self):
        self.v = 0
        self.n = 0

    def u(self, val, n):
        self.v = (self.v << n) | (val & ((1 << n) - 1))
        self.n += n

    def ue(self, val):
        x = val + 1
        k = x.bit_length()
        self.u(0, k - 1)
        self.u(x, k)

    def rbsp(self):
        self.u(1, 1)
        pad = (-self.n) % 8
        self.u(0, pad)
        return self.v.to_bytes(self.n // 8, "big")


def escape(rbsp):
    out = bytearray()
    zeros = 0
    for b in rbsp:
        if zeros >= 2 and b <= 3:
            out.append(3)
            zeros = 0
        out.append(b)
        zeros = zeros + 1 if b == 0 else 0
    return bytes(out)


def dense_cavlc_slice(seed, mbs=8160):
    """SURVEY.md Appendix E 'Dense CAVLC I-slice': per macroblock ue(0)
    I_NxN, 16 x '1' prev_intra4x4_pred_mode_flag, ue(0) chroma pred, ue(0)
    coded_block_pattern code (-> 47), se(0) mb_qp_delta, 16 luma blocks
    '01 s 1', 2 chroma DC '1 s 1', 8 chroma AC '01 s 1': 122 bits."""
    rnd = random.Random(seed)
    w = Bits()
    # slice header for the C1 parameter sets: IDR, nal_ref_idc 3
    w.ue(0)          # first_mb_in_slice
    w.ue(7)          # slice_type I (all slices of the picture)
    w.ue(0)          # pic_parameter_set_id
    w.u(0, 4)        # frame_num
    w.ue(0)          # idr_pic_id
    w.u(0, 2)        # no_output_of_prior_pics, long_term_reference_flag
    w.ue(0)          # slice_qp_delta se(0)
    w.ue(0)          # disable_deblocking_filter_idc
    w.ue(0)          # slice_alpha_c0_offset_div2 se(0)
    w.ue(0)          # slice_beta_offset_div2 se(0)
    assert w.n == 20
    for _ in range(mbs):
        w.ue(0)
        w.u(0xffff, 16)
        w.ue(0)
        w.ue(0)
        w.ue(0)
        for _b in range(16):
            w.u(0b0100 | (rnd.getrandbits(1) << 1) | 1, 4)
        for _b in range(2):
            w.u(0b100 | (rnd.getrandbits(1) << 1) | 1, 3)
        for _b in range(8):
            w.u(0b0100 | (rnd.getrandbits(1) << 1) | 1, 4)
    return b"\x65" + escape(w.rbsp())


def test_dense_cavlc_recipe_known_answer():
    """The survey fed this recipe to the real reference: 122 bits per
    macroblock, 124 444 bytes per 1080p slice, parsed as 8160 I_NxN
    macroblocks."""
    cfg = gen.cfg_1080p()
    nal = dense_cavlc_slice(1)
    assert len(nal) == 124444
    rec = orc.parse_nalus([gen.sps(cfg), gen.pps(cfg), nal])
    assert rec.statuses == [0, 0, 0]
    r = rec.results[0]
    assert r.mb_count == 8160 and r.status == 0
    assert r.hdr_len_bits == 28  # NAL header byte + 20 bits
    mbs = [e for e in rec.event_tuples() if e[0] == 7]
    assert len(mbs) == 8160
    assert [e[1] for e in mbs] == list(range(8160))
    assert {e[2] for e in mbs} == {H264_MB_TYPE_I_NxN}
    t2 = rec.mb_records(0)
    # every luma / chroma block carries exactly one coefficient of +-1
    nz = t2[:, 16:64]
    assert (nz[:, :16] == 1).all()
    assert (nz[:, 16:20] == 1).all() and (nz[:, 32:36] == 1).all()  # Cb, Cr
    assert (t2[:, 1] == 0x2f).all()  # cbp 47


# ---- the reference's own inline bit reader ---------------------------------

REF_BITS = os.path.join(orc.ROOT, "oracle", "_ref", "libref_bits.so")


@pytest.mark.skipif(not os.path.exists(REF_BITS),
                    reason="oracle/_ref not built (reference absent)")
def test_bit_reader_against_reference_inline_reader():
    """oracle/_ref/libref_bits.so is compiled from the reference's
    include/h264/h264_bitstream.h (its static inline h264_bs_read_bits with
    emulation prevention): same bits, same positions, same errors as ours."""
    R = C.CDLL(REF_BITS)
    L = orc.lib()
    sig = [C.c_char_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t,
           C.c_void_p, C.c_void_p]
    R.ref_bs_read_script.argtypes = sig
    L.orc_bs_read_script.argtypes = sig
    assert R.ref_sizeof_bitstream() == REFERENCE_SIZEOF["h264_bitstream"]
    rnd = random.Random(7)
    for case in range(400):
        n = rnd.randrange(0, 200)
        data = bytes(rnd.choice([0, 0, 0, 1, 2, 3, 4, 0x80, 0xff])
                     for _ in range(n))
        k = rnd.randrange(1, 120)
        counts = np.array([rnd.randrange(0, 33) for _ in range(k)],
                          dtype=np.uint8)
        outs = []
        for fn in (R.ref_bs_read_script, L.orc_bs_read_script):
            vals = np.zeros(k, dtype=np.uint32)
            pos = np.zeros(k, dtype=np.uint64)
            done = fn(data, n, case % 2, counts.ctypes.data, k,
                      vals.ctypes.data, pos.ctypes.data)
            outs.append((done, vals.copy(), pos.copy()))
        assert outs[0][0] == outs[1][0], case
        m = min(outs[0][0] + 1, k)
        assert np.array_equal(outs[0][1][:m], outs[1][1][:m]), case
        assert np.array_equal(outs[0][2][:m], outs[1][2][:m]), case


# ---- committed golden fixtures -----------------------------------------------

GOLDEN = os.path.join(orc.ROOT, "tests", "golden")


@pytest.mark.parametrize("name", ["mixed_small", "mixed_400", "mixed_422_10",
                                  "mixed_444", "mixed_interlaced"])
def test_oracle_reproduces_golden_streams(name):
    """tests/golden/*.npz were produced by tests/golden/make_golden.py; the
    oracle must keep producing them (regression pin across rounds)."""
    import golden_io
    fx = golden_io.load(os.path.join(GOLDEN, name))
    assert hashlib.sha256(fx["stream"]).hexdigest() == fx["sha256"]
    rec = orc.parse_stream(fx["stream"], want=orc.ORC_REC_WANT_T2)
    got = golden_io.summarise(rec)
    for key in fx["expect"]:
        assert np.array_equal(got[key], fx["expect"][key]), key


def split_nals(fx):
    sizes = fx["extra"]["nal_sizes"]
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(int)
    return [fx["stream"][offs[i]:offs[i + 1]] for i in range(len(sizes))]


@pytest.mark.parametrize("name", ["damaged_small", "damaged_overflow"])
def test_oracle_reproduces_golden_damaged(name):
    import golden_io
    fx = golden_io.load(os.path.join(GOLDEN, name))
    dropped = C.c_uint64.in_dll(orc.lib(), "orc_stat_dropped_coeffs")
    before = dropped.value
    rec = orc.parse_nalus(split_nals(fx), want=orc.ORC_REC_WANT_T2)
    if name == "damaged_overflow":
        # every slice of this fixture places a coefficient outside its block
        # and the walk goes on, as the reference's does
        # (src/h264_syntax_slice_data.h:222-226)
        assert dropped.value - before >= len(rec.results) == 8
        assert all(r.mb_count >= 4 for r in rec.results)
    assert rec.statuses == list(fx["extra"]["statuses"])
    got = golden_io.summarise(rec)
    for key in ("events", "headers", "results", "mbrecs"):
        assert np.array_equal(got[key], fx["expect"][key]), key


def test_oracle_reproduces_golden_synth_and_cabac():
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "make_golden", os.path.join(GOLDEN, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    z = np.load(os.path.join(GOLDEN, "synth_grid.npz"))
    b = orc.Builder()
    i = 0
    for entropy in (0, 1):
        sps, pps = mg.ps_1080p(entropy)
        for kind, mbc, nal_type, ref, sh in mg.synth_cases(entropy):
            out = b.nalu(ref, nal_type, sps=sps, pps=pps, sh=sh,
                         kind=kind + 1, mb_count=mbc, cap=1 << 16)
            assert len(out) == z["sizes"][i]
            assert hashlib.sha256(out).digest() == z["sha256"][i].tobytes()
            i += 1
    assert i == len(z["sizes"])
    z = np.load(os.path.join(GOLDEN, "cabac_init.npz"))
    for t, d in zip(z["triples"], z["sha256"]):
        st = orc.cabac_init_states(int(t[0]), int(t[1]), int(t[2]))
        assert hashlib.sha256(st.tobytes()).digest() == d.tobytes()
    assert np.array_equal(orc.cabac_init_states(2, 0, 26), z["qp26_I"])
