"""CABAC slice data of REAL streams.

The reference has no CABAC decoder and ships no stream, so the CABAC decoder of
this repository (oracle/orc_cabac_dec.c, kernels/cabac_slice.hip) is written
from the standard and cross-checked only against an encoder by the same hand.
The container image, however, holds two H.264 files from third-party encoders
(sample videos of the `imageio` Python package):

* realshort.mp4 -- High profile 4:2:0, 320 x 240, 36 pictures (I and P), CABAC,
  8x8 transform;
* cockatoo.mp4  -- x264 core 142, High 4:4:4 Predictive, 1280 x 720, 280
  pictures: I, P and B slices, B pyramid, weighted prediction for P and B,
  three reference pictures with mixed references, 8x8 transform, spatial
  direct, adaptive quantisation.

A binary arithmetic decoder whose context modelling, binarisation or neighbour
derivation is off by one anywhere loses such a stream within a few macroblocks
and never finds the end: "every slice yields exactly PicSizeInMbs macroblocks,
status 0, and the decoding engine stops at the rbsp_stop_one_bit" is the test.
cockatoo.mp4 comes from an x264 build older than 151 and carries that
encoder's known 4:4:4 slip (include/h264_amd.h,
H264_AMD_READER_FLAGS_CABAC_X264_PRE151): with the standard's rule a decoder
derails at the first 8x8-transformed macroblock beside one that is not -- this
too is checked, it is what a conforming decoder does with that file.

The files are not copied into the repository (they are not ours); the tests
skip where the image does not have them.  H264_AMD_SAMPLE_DIR overrides the
directory.
"""
import ctypes as C
import glob
import os

import pytest

import mp4_demux
import orc

FLAGS = 0x101  # SLICE_DATA | CABAC


def sample(name):
    cands = []
    if os.environ.get("H264_AMD_SAMPLE_DIR"):
        cands.append(os.path.join(os.environ["H264_AMD_SAMPLE_DIR"], name))
    cands += glob.glob("/opt/conda/lib/python3*/site-packages/imageio/"
                       "resources/images/" + name)
    cands += glob.glob("/usr/lib/python3*/site-packages/imageio/resources/"
                       "images/" + name)
    for c in cands:
        if os.path.isfile(c):
            return c
    pytest.skip("%s is not in this image" % name)


def slices_of(path):
    sps, pps, samples = mp4_demux.video_nals(path)
    nals = [bytes(n) for s in samples for n in s if (n[0] & 31) in (1, 5)]
    return [bytes(x) for x in sps] + [bytes(x) for x in pps], nals


def ends_at_stop_bit(nal, end_bit):
    """end_bit (bits from the start of the NAL unit up to and including the
    rbsp_stop_one_bit, as the slice result reports it): the bit there is a 1
    and lies in the last byte of the RBSP.  (Not "the last 1 of the NAL
    unit": x264 leaves a bit of its own in the alignment padding behind the
    stop bit.)"""
    out, z = bytearray(), 0
    for b in nal[1:]:
        if z >= 2 and b == 3:
            z = 0
            continue
        out.append(b)
        z = z + 1 if b == 0 else 0
    while out and out[-1] == 0:
        out.pop()  # cabac_zero_words
    p = end_bit - 8 - 1
    if p < 0 or p // 8 != len(out) - 1:
        return False
    return (out[p // 8] >> (7 - p % 8)) & 1 == 1


def quirk(on):
    C.c_int.in_dll(orc.lib(), "orc_quirk_x264_444_cbf").value = int(on)


def complete(ref, nals, pic):
    n = 0
    for r, nal in zip(ref.results, nals):
        if (r.status, r.mb_count, r.first_mb_in_slice) == (0, pic, 0) and \
                ends_at_stop_bit(nal, r.end_bit):
            n += 1
    return n


def test_high_profile_stream_decodes_to_the_last_bit():
    head, nals = slices_of(sample("realshort.mp4"))
    ref = orc.parse_nalus(head + nals, flags=FLAGS, want=0)
    sps = ref.sps[0]
    pic = (sps.pic_width_in_mbs_minus1 + 1) * \
        (sps.pic_height_in_map_units_minus1 + 1)
    assert (pic, len(nals)) == (300, 36)
    assert ref.pps[0].entropy_coding_mode_flag == 1
    assert complete(ref, nals, pic) == len(nals)


def test_x264_444_stream_with_b_slices_and_weighted_prediction():
    head, nals = slices_of(sample("cockatoo.mp4"))
    try:
        quirk(True)
        ref = orc.parse_nalus(head + nals, flags=FLAGS, want=0)
    finally:
        quirk(False)
    sps, pps = ref.sps[0], ref.pps[0]
    assert (sps.profile_idc, sps.chroma_format_idc) == (244, 3)
    assert (pps.weighted_pred_flag, pps.weighted_bipred_idc) == (1, 2)
    pic = (sps.pic_width_in_mbs_minus1 + 1) * \
        (sps.pic_height_in_map_units_minus1 + 1)
    assert (pic, len(nals)) == (3600, 280)
    types = sorted(set(h.slice_type % 5 for h in ref.headers))
    assert types == [0, 1, 2]  # P, B, I
    assert complete(ref, nals, pic) == len(nals)
    # the standard's rule on the same file: the first picture is lost behind
    # its first Intra16x16 macroblock (no 8x8 transform) among Intra8x8 ones
    ref = orc.parse_nalus(head + nals[:1], flags=FLAGS, want=0)
    assert ref.results[0].mb_count < pic


def test_x264_build_number_from_the_sei():
    """h264_amd_x264_build (include/h264_amd.h): what a caller keys
    H264_AMD_READER_FLAGS_CABAC_X264_PRE151 to"""
    import ctypes as C
    from libh264_amd import capi
    L = capi.engine_lib()
    L.h264_amd_x264_build.restype = C.c_uint32
    L.h264_amd_x264_build.argtypes = [C.c_char_p, C.c_size_t]
    sps, pps, samples = mp4_demux.video_nals(sample("cockatoo.mp4"))
    sei = [bytes(n) for n in samples[0] if (n[0] & 31) == 6][0]
    assert L.h264_amd_x264_build(sei, len(sei)) == 142
    idr = [bytes(n) for n in samples[0] if (n[0] & 31) == 5][0]
    assert L.h264_amd_x264_build(idr, len(idr)) == 0
    assert L.h264_amd_x264_build(b"\x06\x05x264 - core 164 r3095", 24) == 164
    assert L.h264_amd_x264_build(b"\x06", 1) == 0
