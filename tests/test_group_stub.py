"""The engine group's C entry point (h264_amd_group_decode_slices,
include/h264_amd.h) at world 2 over a STUB device layer: host memory for the
members' "device" memory, the oracle as the per-member decoder.  Everything
between the caller and the decoder is the product's own code -- the plan
(members that get work, size-balanced assignment), the packed pieces, the copy
per member, the per-member descriptors, the records scattered back into the
caller's order.  CPU only (the HIP layer runs in test_gpu_group.py)."""
import ctypes as C

import numpy as np
import pytest

import orc
import streams
from libh264_amd import gen, Params
from libh264_amd.capi import engine_lib
from libh264_amd.ctypes_abi import DecodeArgs, SliceResult, SliceHeader

vp, u32, u64, sz = C.c_void_p, C.c_uint32, C.c_uint64, C.c_size_t


class GroupOps(C.Structure):
    _fields_ = [
        ("ctx", vp),
        ("alloc", C.CFUNCTYPE(C.c_int, vp, u32, sz, C.POINTER(vp))),
        ("free", C.CFUNCTYPE(C.c_int, vp, u32, vp)),
        ("upload", C.CFUNCTYPE(C.c_int, vp, u32, vp, vp, sz)),
        ("download", C.CFUNCTYPE(C.c_int, vp, u32, vp, vp, sz)),
        ("copy", C.CFUNCTYPE(C.c_int, vp, u32, vp, u32, vp, sz)),
        ("pack", C.CFUNCTYPE(C.c_int, vp, u32, vp, vp, vp, vp, vp, u32)),
        ("scatter_rows", C.CFUNCTYPE(C.c_int, vp, u32, vp, vp, vp, u32, sz)),
        ("decode", C.CFUNCTYPE(C.c_int, vp, u32, vp, C.POINTER(DecodeArgs))),
        ("sync", C.CFUNCTYPE(C.c_int, vp, u32)),
    ]


class StubLayer:
    """Members' memories are host buffers; every buffer belongs to the member
    it was allocated on, and a member's decoder may only touch its own."""

    def __init__(self, head_nals):
        self.head = head_nals
        self.owner = {}      # address -> (member, size, keepalive)
        self.log = []
        self.copied = {}     # (dst_member, src_member) -> bytes
        f = GroupOps._fields_
        self.ops = GroupOps()
        for (name, typ), fn in zip(f[1:], (
                self.alloc, self.free, self.upload, self.download, self.copy,
                self.pack, self.scatter_rows, self.decode, self.sync)):
            setattr(self.ops, name, typ(fn))

    def member_of(self, addr, size=1):
        for base, (m, n, _k) in self.owner.items():
            if base <= addr and addr + size <= base + n:
                return m
        raise AssertionError("address %#x outside every allocation" % addr)

    def alloc(self, _ctx, m, size, ret):
        buf = C.create_string_buffer(size + 64)
        self.owner[C.addressof(buf)] = (m, size + 64, buf)
        ret[0] = C.addressof(buf)
        return 0

    def free(self, _ctx, m, p):
        assert self.owner.pop(p)[0] == m
        return 0

    def upload(self, _ctx, m, d, s, n):
        assert self.member_of(d, n) == m
        C.memmove(d, s, n)
        return 0

    def download(self, _ctx, m, d, s, n):
        assert self.member_of(s, n) == m
        C.memmove(d, s, n)
        return 0

    def copy(self, _ctx, dm, d, sm, s, n):
        assert self.member_of(d, n) == dm and self.member_of(s, n) == sm
        C.memmove(d, s, n)
        self.copied[(dm, sm)] = self.copied.get((dm, sm), 0) + n
        return 0

    def pack(self, _ctx, m, d, s, so, do, sizes, n):
        so = (u64 * n).from_address(so)
        do = (u64 * n).from_address(do)
        sizes = (u32 * n).from_address(sizes)
        for i in range(n):
            assert do[i] % 16 == 0
            assert self.member_of(d + do[i], sizes[i]) == m
            C.memmove(d + do[i], s + so[i], sizes[i])
        self.log.append(("pack", m, n))
        return 0

    def scatter_rows(self, _ctx, m, d, s, index, n, row):
        index = (u32 * n).from_address(index)
        for j in range(n):
            C.memmove(d + index[j] * row, s + j * row, row)
        return 0

    def decode(self, _ctx, m, _params, pa):
        a = pa.contents
        descs = np.frombuffer((C.c_uint8 * (16 * a.n)).from_address(a.d_descs),
                              dtype=[("offset", "<u8"), ("size", "<u4"),
                                     ("epoch", "<u4")])
        nals = []
        for d in descs:
            if m != 0 or self.member_of(a.d_buf) != 0:
                assert self.member_of(a.d_buf + int(d["offset"]),
                                      int(d["size"])) == m
            nals.append(C.string_at(a.d_buf + int(d["offset"]), int(d["size"])))
        ref = orc.parse_nalus(self.head + nals, flags=a.reader_flags)
        assert self.member_of(a.d_results, 64 * a.n) == m
        k = 0
        for j in range(a.n):
            C.memmove(a.d_results + 64 * j, C.byref(ref.results[j]), 64)
            if a.d_headers and ref.results[j].status == 0:
                C.memmove(a.d_headers + C.sizeof(SliceHeader) * j,
                          C.byref(ref.headers[k]), C.sizeof(SliceHeader))
            k += ref.results[j].status == 0
            if a.d_mb_out and a.mb_stride:
                rec = ref.mb_records(j)[:a.mb_capacity]
                row = np.zeros((a.mb_capacity, a.mb_stride), dtype=np.uint8)
                row[:len(rec)] = rec[:, :a.mb_stride]
                C.memmove(a.d_mb_out + j * a.mb_capacity * a.mb_stride,
                          row.ctypes.data, row.nbytes)
        self.log.append(("decode", m, a.n))
        return 0

    def sync(self, _ctx, m):
        return 0


def setup_lib():
    L = engine_lib()
    L.h264_amd_group_new_with_ops.argtypes = [C.POINTER(GroupOps), u32,
                                              C.POINTER(vp)]
    L.h264_amd_group_destroy.argtypes = [vp]
    L.h264_amd_group_decode_slices.argtypes = [vp, vp, C.POINTER(DecodeArgs),
                                               vp, C.POINTER(u32)]
    L.h264_amd_group_plan.argtypes = [vp, u32, u32, vp]
    L.h264_amd_group_plan.restype = u32
    return L


def test_plan():
    L = setup_lib()
    rnd = np.random.default_rng(3)
    sizes = rnd.integers(100, 200000, size=6506).astype(np.uint32)
    owner = np.zeros(len(sizes), dtype=np.uint32)
    used = L.h264_amd_group_plan(sizes.ctypes.data, len(sizes), 8,
                                 owner.ctypes.data)
    assert used == 7          # ceil(6506 / 1024): not all eight get work
    load = np.bincount(owner, weights=sizes, minlength=used)
    assert owner.max() == used - 1
    assert load.max() - load.min() <= sizes.max()
    # a 407-slice stream (64 MiB of 1080p) is not scattered at all
    assert L.h264_amd_group_plan(sizes.ctypes.data, 407, 8, owner.ctypes.data) == 1
    assert not owner[:407].any()
    assert L.h264_amd_group_plan(sizes.ctypes.data, 4096, 2, None) == 2
    assert L.h264_amd_group_plan(sizes.ctypes.data, 1025, 8, None) == 2


@pytest.mark.parametrize("n,want_used", [(2300, 2), (300, 1)])
def test_group_decode_over_stub_layer(n, want_used):
    L = setup_lib()
    cfg = streams.small_cfg(width_mbs=4, height_mbs=2)
    head = [gen.sps(cfg), gen.pps(cfg)]
    nals = [gen.slice_nal(cfg, streams.rich_slice(cfg, s, (2, 0, 1)[s % 3],
                                                  density=1 + s % 9))
            for s in range(n)]
    ref = orc.parse_nalus(head + nals)
    params = Params()
    params.put_sps(ref.sps[0])
    epoch = params.put_pps(ref.pps[0])
    layer = StubLayer(head)
    grp = vp()
    assert L.h264_amd_group_new_with_ops(C.byref(layer.ops), 2,
                                         C.byref(grp)) == 0
    # the caller's buffers live on member 0
    buf, offs, sizes = streams.pack_nals(nals, lead=1)
    def dev0(nbytes):
        p = vp()
        layer.alloc(None, 0, nbytes, C.pointer(p))
        return p.value
    d_buf = dev0(len(buf))
    C.memmove(d_buf, buf.ctypes.data, len(buf))
    descs = np.zeros(n, dtype=[("offset", "<u8"), ("size", "<u4"),
                               ("epoch", "<u4")])
    descs["offset"], descs["size"], descs["epoch"] = offs, sizes, epoch
    d_descs = dev0(descs.nbytes)
    C.memmove(d_descs, descs.ctypes.data, descs.nbytes)
    cap, stride = 8, 16
    a = DecodeArgs()
    a.d_buf, a.d_descs, a.n = d_buf, d_descs, n
    a.reader_flags, a.tier, a.mb_capacity, a.mb_stride = 1, 1, cap, stride
    a.d_results = dev0(64 * n)
    a.d_headers = dev0(C.sizeof(SliceHeader) * n)
    a.d_mb_out = dev0(n * cap * stride)
    a.d_buf_len = len(buf)
    used = u32()
    for step in range(2):       # the second call reuses the plan
        layer.log.clear()
        assert L.h264_amd_group_decode_slices(grp, params.h, C.byref(a), None,
                                              C.byref(used)) == 0
        assert used.value == want_used
        got = (SliceResult * n).from_address(a.d_results)
        for i in range(n):
            assert bytes(got[i]) == bytes(ref.results[i]), i
            hdr = C.string_at(a.d_headers + C.sizeof(SliceHeader) * i,
                              C.sizeof(SliceHeader))
            assert hdr == bytes(ref.headers[i]), i
            mb = np.frombuffer(C.string_at(a.d_mb_out + i * cap * stride,
                                           cap * stride),
                               dtype=np.uint8).reshape(cap, stride)
            assert np.array_equal(mb, ref.mb_records(i)[:cap, :stride]), i
        dec = [e for e in layer.log if e[0] == "decode"]
        if want_used == 1:
            assert dec == [("decode", 0, n)]
            assert not layer.copied
        else:
            assert sorted(e[1] for e in dec) == [0, 1]
            assert sum(e[2] for e in dec) == n
            # slice bytes went point to point, member 0 -> member 1, once per
            # step; only records came back
            assert layer.copied[(1, 0)] > 0 and layer.copied[(0, 1)] > 0
    assert L.h264_amd_group_destroy(grp) == 0
