"""The constant tables of libh264_amd/csrc/common/*.inc (H.264 Tables 9-4,
9-5, 9-7..9-10, 9-12..9-33, 9-44, 9-45).

* Structure checks that need nothing but the tables: every VLC class is a
  prefix-free code, complete where the standard's is, and covers each symbol
  exactly once.
* In the development container, where the reference's text exists, the tuples
  are re-derived from its restatement of the same tables
  (tools/transcribe_spec_tables.py) and must equal the committed files.  This
  test is skipped on the GPU box (no /root/reference there)."""
import importlib.util
import os
import re
from collections import defaultdict
from fractions import Fraction

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = os.path.join(ROOT, "libh264_amd", "csrc", "common")


def rows(fname, macro):
    pat = re.compile(r"^%s\((.*)\)\s*$" % macro)
    out = []
    for line in open(os.path.join(COMMON, fname)):
        m = pat.match(line)
        if m:
            out.append([a.strip() for a in m.group(1).split(",")])
    return out


def code(a):
    return a.strip('"').replace(" ", "")


def check_prefix_free(codes):
    s = sorted(codes)
    for a, b in zip(s, s[1:]):
        assert not b.startswith(a), (a, b)
    return sum(Fraction(1, 2 ** len(c)) for c in codes)


def test_coeff_token_tables():
    per_col = defaultdict(list)
    for col, bits, t1, tc in rows("h264_cavlc_tables.inc", "H264_CT"):
        per_col[int(col)].append((code(bits), int(t1), int(tc)))
    assert sorted(per_col) == [0, 1, 2, 3, 4, 5]
    for col, ents in per_col.items():
        max_tc = {4: 4, 5: 8}.get(col, 16)
        want = {(t1, tc) for tc in range(max_tc + 1)
                for t1 in range(min(tc, 3) + 1)}
        assert {(t1, tc) for _, t1, tc in ents} == want, col
        assert len(ents) == len(want)
        kraft = check_prefix_free([c for c, _, _ in ents])
        assert kraft <= 1
        if col == 3:
            assert all(len(c) == 6 for c, _, _ in ents)
    # spot values printed in Table 9-5
    first = {(t1, tc): c for c, t1, tc in per_col[0]}
    assert first[(0, 0)] == "1" and first[(1, 1)] == "01"
    assert first[(3, 3)] == "00011" and first[(0, 16)] == "0000000000000100"


def test_total_zeros_and_run_before():
    for macro, nidx, maxv in (("H264_TZ", 15, None), ("H264_TZ_C420", 3, None),
                              ("H264_TZ_C422", 7, None), ("H264_RB", 7, None)):
        per = defaultdict(list)
        for idx, bits, v in rows("h264_cavlc_tables.inc", macro):
            per[int(idx)].append((code(bits), int(v)))
        assert len(per) == nidx, macro
        for idx, ents in per.items():
            kraft = check_prefix_free([c for c, _ in ents])
            vals = sorted(v for _, v in ents)
            assert vals == list(range(len(vals))), (macro, idx)
            if macro == "H264_TZ":
                # tzVlcIndex i (1-based) codes total_zeros 0..16-i
                assert len(vals) == 16 - idx + 1 or len(vals) == 17 - idx
            if macro == "H264_RB" and idx < 7:
                assert kraft == 1, (macro, idx)
    rb7 = [c for i, c, v in rows("h264_cavlc_tables.inc", "H264_RB")
           if int(i) == 7 or int(i) == max(int(r[0]) for r in
                                           rows("h264_cavlc_tables.inc",
                                                "H264_RB"))]
    assert any(code(c) == "00000000001" for c in rb7)  # run_before 14


def test_cbp_mapping_is_a_permutation():
    per = defaultdict(list)
    for cls, codenum, intra, inter in rows("h264_cavlc_tables.inc",
                                           "H264_CBP"):
        per[int(cls)].append((int(codenum), int(intra), int(inter)))
    assert sorted(per) == [0, 1]
    for cls, ents in per.items():
        n = 48 if cls == 0 else 16
        assert sorted(e[0] for e in ents) == list(range(n))
        for col in (1, 2):
            vals = sorted(e[col] for e in ents)
            assert vals == (list(range(48)) if cls == 0 else
                            list(range(16))), (cls, col)
    m0 = {c: (a, b) for c, a, b in per[0]}
    assert m0[0] == (47, 0) and m0[1] == (31, 16)  # Table 9-4, first rows


def test_cabac_constants():
    lps = rows("h264_cabac_tables.inc", "H264_LPS")
    assert len(lps) == 64
    assert [int(x) for x in lps[0][1:]] == [128, 176, 208, 240]
    assert [int(x) for x in lps[63][1:]] == [2, 2, 2, 2]
    for a, b in zip(lps, lps[1:]):
        assert all(int(x) >= int(y) for x, y in zip(a[1:], b[1:]))
    tr = rows("h264_cabac_tables.inc", "H264_TRANS")
    assert len(tr) == 64
    for i, l, m in tr:
        i, l, m = int(i), int(l), int(m)
        assert m == min(i + 1, 62) or (i == 63 and m == 63)
        assert l <= i
    mn = rows("h264_cabac_tables.inc", "H264_MN")
    assert len(mn) == 1024
    d = {int(r[0]): [int(x) for x in r[1:]] for r in mn}
    assert d[0][:2] == [20, -15]        # Table 9-12, ctxIdx 0
    assert d[276] == [0] * 8            # terminate context has no (m, n)
    assert d[11][2:4] == [23, 33]       # Table 9-13, ctxIdx 11 idc 0


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"),
                    reason="reference text not present (GPU box)")
def test_tables_equal_reference_text(tmp_path, monkeypatch):
    spec = importlib.util.spec_from_file_location(
        "transcribe", os.path.join(ROOT, "tools",
                                   "transcribe_spec_tables.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    monkeypatch.setattr(mod, "OUT", tmp_path)
    mod.emit_cavlc()
    mod.emit_cabac()
    for f in ("h264_cavlc_tables.inc", "h264_cabac_tables.inc"):
        assert (tmp_path / f).read_text() == open(
            os.path.join(COMMON, f)).read(), f
