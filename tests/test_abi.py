"""The C-ABI boundary (CPU): both shared libraries load, export every function
include/*.h declares, and the public struct layouts have the sizes measured on
the reference's headers.  No compute entry point is called."""
import ctypes as C
import os
import re
import subprocess

import pytest

from h264types import REFERENCE_SIZEOF

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include")
LIBDIR = os.path.join(ROOT, "libh264_amd", "lib")

DECL = re.compile(r"H264_API\s+[^;()]*?\b(h264_\w+)\s*\(", re.S)


def declared(path):
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(DECL.findall(text)))


@pytest.fixture(scope="module")
def libs():
    if not os.path.exists(os.path.join(LIBDIR, "libh264.so")):
        from libh264_amd import build
        build.build_all()
    eng = C.CDLL(os.path.join(LIBDIR, "libh264_amd_engine.so"),
                 mode=C.RTLD_GLOBAL)
    host = C.CDLL(os.path.join(LIBDIR, "libh264.so"))
    return eng, host


def test_engine_exports_every_declared_symbol(libs):
    eng, _ = libs
    names = declared(os.path.join(INC, "h264_amd.h"))
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(eng, n)]
    assert missing == []


def test_dropin_exports_every_declared_symbol(libs):
    _, host = libs
    names = []
    for f in sorted(os.listdir(os.path.join(INC, "h264"))):
        names += declared(os.path.join(INC, "h264", f))
    # the reference's public surface: ctx, reader, writer, bitstream, types
    for must in ("h264_reader_new", "h264_reader_parse",
                 "h264_reader_parse_nalu", "h264_reader_stop",
                 "h264_write_nalu", "h264_write_grey_i_slice",
                 "h264_write_skipped_p_slice", "h264_rewrite_slice_header",
                 "h264_ctx_new", "h264_ctx_set_sps", "h264_ctx_add_sei",
                 "h264_find_nalu", "h264_bs_write_bits", "h264_get_info",
                 "h264_parse_sps", "h264_parse_pps",
                 "h264_byte_stream_to_avcc", "h264_avcc_to_byte_stream",
                 "h264_sar_to_aspect_ratio_idc"):
        assert must in names, must
    missing = [n for n in names if not hasattr(host, n)]
    assert missing == []


def test_dropin_hides_private_symbols():
    out = subprocess.check_output(
        ["nm", "-D", "--defined-only", os.path.join(LIBDIR, "libh264.so")],
        text=True)
    syms = [l.split()[-1] for l in out.splitlines() if l.strip()]
    leaked = [s for s in syms if s.startswith("h264_host_") or
              s.startswith("h264_syn_")]
    assert leaked == []


def test_struct_sizes_with_the_product_headers(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text("""
#include <stdio.h>
#include <h264/h264.h>
#include <h264_amd.h>
#define P(t) printf(#t " %zu\\n", sizeof(struct t))
int main(void) {
  P(h264_bitstream); P(h264_nalu_header); P(h264_sps); P(h264_pps);
  P(h264_slice_header); P(h264_sei); P(h264_sps_derived); P(h264_info);
  P(h264_ctx_cbs); P(h264_amd_slice_result); P(h264_amd_mb_summary);
  P(h264_amd_nal); P(h264_amd_slice_desc);
  return 0; }
""")
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-std=gnu99", "-I" + INC, str(src), "-o",
                           str(exe)])
    got = dict(l.split() for l in
               subprocess.check_output([str(exe)], text=True).splitlines())
    for name, size in REFERENCE_SIZEOF.items():
        assert int(got[name]) == size, name
    assert int(got["h264_amd_slice_result"]) == 64
    assert int(got["h264_amd_mb_summary"]) == 64
    assert int(got["h264_amd_nal"]) == 16
    assert int(got["h264_amd_slice_desc"]) == 16


def test_headers_compile_as_cxx(tmp_path):
    src = tmp_path / "x.cpp"
    src.write_text("#include <h264/h264.h>\n#include <h264_amd.h>\n"
                   "int main() { return 0; }\n")
    subprocess.check_call(["g++", "-I" + INC, "-fsyntax-only", str(src)])


@pytest.mark.skipif(not os.path.isdir("/root/reference/include/h264"),
                    reason="reference headers not present (GPU box)")
def test_public_surface_equals_the_reference():
    """Every function the reference declares in include/h264/*.h is declared
    here too -- the JSON dump included (own JSON tree instead of json-c) --
    and nothing extra hides in the drop-in headers; every one of them is an
    exported symbol of libh264.so."""
    import glob

    def names(paths):
        out = set()
        for p in paths:
            t = re.sub(r"/\*.*?\*/", "", open(p).read(), flags=re.S)
            out |= set(re.findall(r"\b(h264_\w+)\s*\(", t))
        return out
    ref = names(glob.glob("/root/reference/include/h264/*.h"))
    mine = names(glob.glob(os.path.join(INC, "h264", "*.h")))
    assert sorted(ref - mine) == []
    assert sorted(mine - ref) == []
    # static inline helpers of h264_bitstream.h are ABI by inclusion, the
    # rest must be exported
    lib = os.path.join(ROOT, "libh264_amd", "lib", "libh264.so")
    exported = set(subprocess.check_output(
        ["nm", "-D", "--defined-only", lib], text=True).split())
    inline = names([os.path.join(INC, "h264", "h264_bitstream.h")])
    missing = sorted(n for n in ref if n not in exported and n not in inline)
    assert missing == []
    assert len([n for n in ref if n in exported]) >= 64


def _layout(tmp_path, include_dir, tag):
    exe = tmp_path / ("lp_" + tag)
    subprocess.check_call(["gcc", "-std=gnu99", "-I" + include_dir,
                           os.path.join(ROOT, "tests", "helpers",
                                        "layout_probe.c"), "-o", str(exe)])
    return subprocess.check_output([str(exe)], text=True)


def test_struct_layout_field_by_field(tmp_path):
    """99 sizeof / offsetof / enum probes of the public structs.  The expected
    values (tests/golden/public_layout.txt) were printed by the same probe
    compiled against the REFERENCE's headers; the drop-in's headers must give
    the same, and so must the reference's whenever it is present."""
    want = open(os.path.join(ROOT, "tests", "golden",
                             "public_layout.txt")).read()
    assert _layout(tmp_path, INC, "mine") == want
    if os.path.isdir("/root/reference/include/h264"):
        assert _layout(tmp_path, "/root/reference/include", "ref") == want
