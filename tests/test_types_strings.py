"""h264_*_str(): names derived from the reference's own text (its
H264_ENUM_CASE lists in src/h264_types.c and enum values in
include/h264/h264_types.h) when that text is present; spot values otherwise."""
import ctypes as C
import os
import re

import pytest

import prod

REF = "/root/reference"
FUNCS = (("h264_nalu_type_str", "H264_NALU_TYPE_", "UNKNOWN"),
         ("h264_slice_type_str", "H264_SLICE_TYPE_", "UNKNOWN"),
         ("h264_mb_type_str", "H264_MB_TYPE_", "UNKNOWN"),
         ("h264_profile_str", "H264_PROFILE_", "UNKNOWN"),
         ("h264_color_format_str", "H264_COLOR_FORMAT_", "UNKNOWN"),
         # sic: the reference misspells this one (src/h264_types.c:305)
         ("h264_sei_type_str", "H264_SEI_TYPE_", "UNKNONW"))


def fn(name):
    f = getattr(prod.h264(), name)
    f.restype = C.c_char_p
    f.argtypes = [C.c_int]
    return f


def test_spot_values():
    assert fn("h264_nalu_type_str")(5) == b"SLICE_IDR"
    assert fn("h264_slice_type_str")(2) == b"I"
    assert fn("h264_mb_type_str")(1) == b"I_NxN"
    assert fn("h264_sei_type_str")(6) == b"RECOVERY_POINT"
    for name, _, unknown in FUNCS:
        assert fn(name)(99999) == unknown.encode()


@pytest.mark.skipif(not os.path.isdir(REF + "/src"),
                    reason="reference text not present (GPU box)")
def test_every_name_of_the_reference():
    src = open(REF + "/src/h264_types.c").read()
    hdr = open(REF + "/include/h264/h264_types.h").read()
    enums = {}
    for m in re.finditer(r"enum\s+h264_\w+\s*\{(.*?)\};", hdr, re.S):
        body = re.sub(r"/\*.*?\*/", "", m.group(1), flags=re.S)
        v = -1
        for item in body.split(","):
            item = item.strip()
            if not item:
                continue
            if "=" in item:
                n, e = (x.strip() for x in item.split("="))
                v = int(e, 0) if re.fullmatch(r"-?(0x)?[0-9a-fA-F]+", e) \
                    else enums[e]
            else:
                n = item
                v += 1
            enums[n] = v
    total = 0
    for name, prefix, _ in FUNCS:
        body = re.search(r"const char \*%s\(.*?\n\}" % name, src, re.S).group(0)
        for n in re.findall(r"H264_ENUM_CASE\(%s, (\w+)\)" % prefix, body):
            assert fn(name)(enums[prefix + n]) == n.encode(), (name, n)
            total += 1
    assert total >= 90
