"""The engine group on the HIP device layer (include/h264_amd.h, "engine group"):
pack kernel, one device-to-device copy per member, per-member decode, records
scattered back -- rehearsed on the one GPU of the box with two engines on the
same device (H264_AMD_GROUP_REHEARSE=1; on a multi-GPU host the same code runs
with one engine per device and hipMemcpyPeerAsync over xGMI).  The CPU twin
over a stub layer: test_group_stub.py."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import orc
import streams
from libh264_amd import gen
from libh264_amd.capi import engine_lib
from libh264_amd.ctypes_abi import (DecodeArgs, SliceHeader, TIER_T2,
                                    t2_stride)
from test_gpu_decode import make_params

pytestmark = pytest.mark.gpu
vp, u32 = C.c_void_p, C.c_uint32


def test_group_of_two_engines_equals_one_engine(engine, monkeypatch):
    monkeypatch.setenv("H264_AMD_GROUP_REHEARSE", "1")
    L = engine_lib()
    L.h264_amd_group_new.argtypes = [C.POINTER(C.c_int), u32, C.POINTER(vp)]
    L.h264_amd_group_destroy.argtypes = [vp]
    L.h264_amd_group_decode_slices.argtypes = [vp, vp, C.POINTER(DecodeArgs),
                                               vp, C.POINTER(u32)]
    n = 2300
    cfg = streams.small_cfg(width_mbs=4, height_mbs=2)
    head = [gen.sps(cfg), gen.pps(cfg)]
    nals = [gen.slice_nal(cfg, streams.rich_slice(cfg, s, (2, 0, 1)[s % 3],
                                                  density=1 + s % 9))
            for s in range(n)]
    ref = orc.parse_nalus(head + nals, want=0)
    params = make_params(ref)
    buf, offs, sizes = streams.pack_nals(nals, lead=3)
    d_buf = engine.upload(buf)
    descs = [(o, s, params.epoch) for o, s in zip(offs, sizes)]
    one = engine.decode_slices(params, d_buf, descs, reader_flags=1,
                               tier=TIER_T2, mb_capacity=8,
                               want_headers=True)
    want = {k: engine.download(one[k]).copy()
            for k in ("results", "headers", "mb")}

    devs = (C.c_int * 2)(0, 0)
    grp = vp()
    assert L.h264_amd_group_new(devs, 2, C.byref(grp)) == 0
    try:
        # fresh output buffers on the first member's device (= this engine's)
        out = {k: engine.alloc(one[k].nbytes) for k in ("results", "headers",
                                                        "mb")}
        a = DecodeArgs()
        a.d_buf, a.d_descs, a.n = d_buf.ptr, one["d_descs"].ptr, n
        a.reader_flags, a.tier, a.mb_capacity = 1, TIER_T2, 8
        a.mb_stride = one["stride"]
        a.d_results, a.d_headers = out["results"].ptr, out["headers"].ptr
        a.d_mb_out = out["mb"].ptr
        a.d_buf_len = d_buf.nbytes
        used = u32()
        for _ in range(2):      # the second call reuses plan and buffers
            for k in out:
                engine.memset(out[k], 0xee)
            assert L.h264_amd_group_decode_slices(grp, params.h, C.byref(a),
                                                  None, C.byref(used)) == 0
            assert used.value == 2
            for k in out:
                got = engine.download(out[k])
                assert np.array_equal(got[:len(want[k])], want[k]), k
        # a small batch stays on the first member
        a.n = 300
        assert L.h264_amd_group_decode_slices(grp, params.h, C.byref(a), None,
                                              C.byref(used)) == 0
        assert used.value == 1
    finally:
        assert L.h264_amd_group_destroy(grp) == 0


def test_dropin_reader_over_a_group():
    """libh264.so under H264_AMD_DEVICES=0,0: the reader's chunk goes through
    h264_amd_group_decode_slices; the callback log equals the oracle's."""
    code = r'''
import sys, os
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import orc, prod, streams
from libh264_amd import gen
cfg = streams.small_cfg(width_mbs=4, height_mbs=2)
nals = [gen.sps(cfg), gen.pps(cfg)]
nals += [gen.slice_nal(cfg, streams.rich_slice(cfg, s, (2, 0, 1)[s % 3],
                                               density=1 + s % 5))
         for s in range(2200)]
data = streams.annexb(nals)
ref = orc.parse_stream(data, flags=1, want=0)
got = prod.parse_stream(data, flags=1)
want = [e for e in ref.event_tuples() if e[0] != 13]
assert got.result == 0 and got.event_tuples() == want
print("OK", len(want))
'''
    env = dict(os.environ, H264_AMD_DEVICES="0,0", H264_AMD_GROUP_REHEARSE="1")
    p = subprocess.run([sys.executable, "-c", code], env=env, cwd=orc.ROOT,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and p.stdout.startswith("OK"), p.stderr[-2000:]
