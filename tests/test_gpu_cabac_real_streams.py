"""GPU leg of tests/test_cabac_real_streams.py: the CABAC kernel on the two
third-party streams the image holds, through the C-ABI, against the oracle's
decoder -- every slice to its last macroblock and to the stop bit, checksums
over the full syntax equal."""
import pytest

from libh264_amd.ctypes_abi import (READER_FLAGS_CABAC,
                                    READER_FLAGS_CABAC_X264_PRE151, TIER_T0,
                                    TIER_T2)
from test_cabac_real_streams import (ends_at_stop_bit, quirk, sample,
                                     slices_of)
from test_gpu_cabac import compare

pytestmark = pytest.mark.gpu
FLAGS = 1 | READER_FLAGS_CABAC


def test_high_profile_stream(engine):
    head, nals = slices_of(sample("realshort.mp4"))
    ref, res = compare(engine, head, nals, tier=TIER_T2, cap=302, flags=FLAGS)
    assert [(r.status, r.mb_count) for r in res] == [(0, 300)] * 36
    assert all(ends_at_stop_bit(n, r.end_bit) for n, r in zip(nals, res))


def test_x264_444_stream(engine):
    head, nals = slices_of(sample("cockatoo.mp4"))
    # as the standard has it: lost in the first picture, kernel and oracle at
    # the same macroblock
    ref, res = compare(engine, head, nals[:8], tier=TIER_T0, cap=3602,
                       flags=FLAGS)
    assert res[0].mb_count < 3600
    # as x264 before build 151 wrote it
    try:
        quirk(True)
        ref, res = compare(engine, head, nals, tier=TIER_T0, cap=3602,
                           flags=FLAGS | READER_FLAGS_CABAC_X264_PRE151)
    finally:
        quirk(False)
    assert [(r.status, r.mb_count) for r in res] == [(0, 3600)] * 280
    assert all(ends_at_stop_bit(n, r.end_bit) for n, r in zip(nals, res))
