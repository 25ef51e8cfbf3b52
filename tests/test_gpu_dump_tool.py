"""tools/h264_dump/h264-dump (BASELINE config 1): this repository's dump tool over
the drop-in libh264.so, behaviour of the reference's tools/h264_dump.c:195-263,
309-435 -- a JSON line and "OK" (write-back equals the input) per NAL unit."""
import json
import os
import subprocess

import pytest

import golden_io
import orc
import streams
from libh264_amd import build, gen
from libh264_amd.ctypes_abi import GenSlice

pytestmark = pytest.mark.gpu


def run_tool(path, *extra):
    tool = build.build_tools()
    assert tool and os.path.exists(tool)
    p = subprocess.run([tool] + list(extra) + [str(path)],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    return p


def check_output(text, n_nals):
    lines = text.splitlines()
    assert len(lines) == 2 * n_nals
    docs = []
    for k in range(n_nals):
        docs.append(json.loads(lines[2 * k]))
        assert lines[2 * k + 1] == "OK", k
    return docs


def test_mixed_small(tmp_path):
    g = golden_io.load(os.path.join(os.path.dirname(__file__), "golden",
                                    "mixed_small"))
    f = tmp_path / "mixed_small.h264"
    f.write_bytes(g["stream"])
    n = len(g["expect"]["nals"])
    docs = check_output(run_tool(f).stdout, n)
    assert sum("slice_header" in json.dumps(d) or "first_mb_in_slice" in
               json.dumps(d) for d in docs) > 0
    # -o and --pretty: the same documents, one member per line
    out = tmp_path / "out.json"
    run_tool(f, "-o", out, "--pretty")
    text = out.read_text()
    dec = json.JSONDecoder()
    pos, got = 0, []
    while pos < len(text):
        while pos < len(text) and text[pos] in " \n":
            pos += 1
        if pos >= len(text):
            break
        if text.startswith("OK", pos):
            pos += 2
            continue
        obj, pos = dec.raw_decode(text, pos)
        got.append(obj)
    assert got == docs


def test_1080p_file(tmp_path):
    cfg = gen.cfg_1080p()
    s = GenSlice(slice_type=7, idr=1, nal_ref_idc=3, num_mbs=8160,
                 pm_i16x16=280, pm_pcm=2, density=5, pm_escape=5,
                 pm_cbp_zero=50, seed=17)
    nals = [gen.sps(cfg), gen.pps(cfg), gen.slice_nal(cfg, s)]
    f = tmp_path / "c1.h264"
    f.write_bytes(streams.annexb(nals))
    docs = check_output(run_tool(f).stdout, 3)
    text = json.dumps(docs[2])
    assert text.count("mb_addr") == 8160
    # the reader-side variant (the reference tool's "#if 0" build) and timing
    p = run_tool(f, "--reader-slice-data", "--bench", "2")
    r = json.loads([l for l in p.stderr.splitlines() if l.startswith("{")][-1])
    assert (r["ok"], r["mismatches"], r["slice_data_mb_callbacks"]) == \
        (3, 0, 8160)
