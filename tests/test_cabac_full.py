"""CABAC slice data beyond the reference's writers (spec-only, parity unpinned):
the generator's arithmetic ENCODER for the whole macroblock layer
(libh264_amd/csrc/gen/h264gen_cabac.inc, written from the standard, neighbour
data by position) against the oracle's DECODER (oracle/orc_cabac_dec.c,
neighbour derivations of 6.4.11 by block index).  Every field of every
macroblock record must come back: I / SI / P / SP / B slices, four chroma
formats, high bit depths, 8x8 transform, I_PCM, field pictures, partial
slices, all cabac_init_idc, the QP range.  CPU only; the kernel is compared
with this oracle in test_gpu_cabac_full.py."""
import numpy as np
import pytest

import orc
import streams
from libh264_amd import gen

FLAGS = 1 | 0x100


def cabac_cfg(**kw):
    return streams.small_cfg(entropy_coding_mode_flag=1, **kw)


def cabac_slice(cfg, seed, st, first=0, n=None, **kw):
    kw.setdefault("cabac_init_idc", seed % 3)
    kw.setdefault("cabac_syntax", 1)
    return streams.rich_slice(cfg, seed, st, first, n, **kw)


STD_QP = 0x400   # H264_AMD_READER_FLAGS_CABAC_STD_QP


def roundtrip(cfg, slices, flags=FLAGS):
    nals, expects = [], []
    for s in slices:
        nal, exp = gen.slice_nal(cfg, s, want_t2=True)
        nals.append(nal)
        expects.append(exp)
    rec = orc.parse_nalus([gen.sps(cfg), gen.pps(cfg)] + nals, flags=flags)
    for i, (s, exp) in enumerate(zip(slices, expects)):
        r = rec.results[i]
        assert (r.status, r.mb_count) == (0, s.num_mbs), (i, r.status)
        assert r.flags == 7
        got = rec.mb_records(i)
        if not np.array_equal(got, exp):
            bad = np.argwhere(got != exp)[0]
            raise AssertionError("slice %d mb %d byte %d: %d != %d" % (
                i, bad[0], bad[1], got[bad[0], bad[1]], exp[bad[0], bad[1]]))
    return rec, nals


@pytest.mark.parametrize("cf,bd", [(1, 0), (0, 0), (2, 0), (2, 2), (3, 0),
                                   (3, 4), (1, 6)])
@pytest.mark.parametrize("st", [2, 0, 1])
def test_chroma_formats_and_slice_types(cf, bd, st):
    cfg = cabac_cfg(cf=cf, bd=bd)
    roundtrip(cfg, [cabac_slice(cfg, seed, st) for seed in range(8)])


@pytest.mark.parametrize("st", [2, 0, 1, 3, 4])
def test_field_pictures_and_switching_slices(st):
    cfg = cabac_cfg(field_coding=1)
    roundtrip(cfg, [cabac_slice(cfg, seed, st, field_pic=seed % 3)
                    for seed in range(9)])


def test_dense_444_with_8x8_and_escapes():
    cfg = cabac_cfg(cf=3, bd=4)
    roundtrip(cfg, [cabac_slice(cfg, seed, (2, 0, 1)[seed % 3], pm_t8x8=800,
                                density=12, pm_escape=200)
                    for seed in range(12)])


def test_pcm_restarts_the_arithmetic_decoder():
    cfg = cabac_cfg()
    rec, nals = roundtrip(cfg, [cabac_slice(cfg, seed, (2, 0, 1)[seed % 3],
                                            pm_pcm=300, pm_zero_pcm=900)
                                for seed in range(12)])
    assert any(b"\x00\x00\x03" in n for n in nals)
    assert any((rec.mb_records(i)[:, 0] == 3).any() for i in range(12))


def test_partial_slices_sizes_qp():
    for seed in range(60):
        cf = (1, 0, 2, 3)[seed % 4]
        cfg = cabac_cfg(cf=cf, bd=(0, 2)[seed % 2] if cf != 1 else 0,
                        width_mbs=3 + seed % 7, height_mbs=2 + seed % 5)
        total = cfg.width_mbs * cfg.height_mbs
        first = (seed * 5) % (total - 1)
        n = 1 + (seed * 7) % (total - first)
        roundtrip(cfg, [cabac_slice(cfg, seed * 3 + k,
                                    (2, 0, 1, 3, 4)[(seed + k) % 5], first, n,
                                    slice_qp_delta=(seed % 52) - 26)
                        for k in range(3)])


def qp0_slices(cfg, syntax):
    return [cabac_slice(cfg, 700 + k, (2, 0, 1)[k % 3], slice_qp_delta=-26,
                        cabac_syntax=syntax) for k in range(9)]


def test_slice_qp_0_context_init_of_the_standard():
    """9.3.1.1 clips SliceQPY to 0..51, the reference's encoder to 1..51
    (src/h264_bac.c:216-230): a slice at SliceQPY 0 decodes with the clip its
    encoder used (H264_AMD_READER_FLAGS_CABAC_STD_QP for the standard's), and
    is lost with the other one."""
    cfg = cabac_cfg(cf=1)
    roundtrip(cfg, qp0_slices(cfg, 1))
    roundtrip(cfg, qp0_slices(cfg, 2), flags=FLAGS | STD_QP)
    for syntax, flags in ((2, FLAGS), (1, FLAGS | STD_QP)):
        with pytest.raises(AssertionError):
            roundtrip(cfg, qp0_slices(cfg, syntax), flags=flags)
    # above QP 0 the flag changes nothing
    cfg = cabac_cfg(cf=3)
    s = [cabac_slice(cfg, 720 + k, (2, 0, 1)[k % 3], slice_qp_delta=-25 + k,
                     cabac_syntax=1 + k % 2) for k in range(6)]
    roundtrip(cfg, s)
    roundtrip(cfg, s, flags=FLAGS | STD_QP)


def test_without_the_flag_nothing_is_walked():
    """the reference's behaviour (src/h264_syntax_slice_data.h:715-717)"""
    cfg = cabac_cfg()
    nal = gen.slice_nal(cfg, cabac_slice(cfg, 3, 0))
    r = orc.parse_nalus([gen.sps(cfg), gen.pps(cfg), nal], flags=1).results[0]
    assert (r.status, r.mb_count, r.flags) == (0, 0, 5)


def test_1080p_p_slice():
    cfg = gen.cfg_1080p(profile_idc=100, entropy_coding_mode_flag=1,
                        transform_8x8_mode_flag=1)
    s = gen.GenSlice(slice_type=5, nal_ref_idc=2, num_mbs=8160, density=5,
                     pm_i16x16=300, pm_skip=250, pm_intra_in_inter=60,
                     pm_t8x8=300, pm_escape=5, pm_cbp_zero=150,
                     cabac_syntax=1, seed=99)
    rec, nals = roundtrip(cfg, [s])
    assert len(nals[0]) > 60000


def test_damaged_slices_end_defined():
    """bit flips and truncation: a status, never a crash; macroblocks delivered
    before the error are whole records"""
    import random
    rnd = random.Random(5)
    cfg = cabac_cfg()
    head = [gen.sps(cfg), gen.pps(cfg)]
    for seed in range(40):
        nal = bytearray(gen.slice_nal(
            cfg, cabac_slice(cfg, seed, (2, 0, 1)[seed % 3])))
        if seed % 2:
            del nal[rnd.randrange(6, len(nal)):]
        else:
            for _ in range(1 + seed % 3):
                p = rnd.randrange(4, len(nal))
                nal[p] ^= 1 << rnd.randrange(8)
        r = orc.parse_nalus(head + [bytes(nal)], flags=FLAGS).results[0]
        assert r.status in (0, -5, -22, -38, -75, -2)


def test_illegal_start_of_the_decoding_engine():
    """codIOffset 510 / 511 at the start (9.3.1.2 forbids them): -EIO, not
    arithmetic on a state the engine is not defined for (found by
    tools/fuzz_cabac_full.py: oracle and kernel diverged there)"""
    cfg = cabac_cfg()
    nal = gen.slice_nal(cfg, cabac_slice(cfg, 3, 0))
    r = orc.parse_nalus([gen.sps(cfg), gen.pps(cfg), nal],
                        flags=FLAGS).results[0]
    assert r.status == 0
    bad = bytearray(nal)
    bad[r.rawdata_off] = bad[r.rawdata_off + 1] = 0xff
    r2 = orc.parse_nalus([gen.sps(cfg), gen.pps(cfg), bytes(bad)],
                         flags=FLAGS).results[0]
    assert (r2.status, r2.mb_count) == (-5, 0)
