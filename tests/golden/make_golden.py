#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz.

Inputs come from the seeded generator (libh264_amd/csrc/gen, an independent
CAVLC encoder); expected outputs from the oracle (oracle/, the CPU restatement
of the reference's parse/write paths, pinned as described in
tests/test_oracle_pins.py).  The reference ships no fixtures of its own and
does not build in this image, so these files are what carries the expected
behaviour to the GPU box and from round to round.

    python tests/golden/make_golden.py
"""
import ctypes as C
import hashlib
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import golden_io  # noqa: E402
import orc  # noqa: E402
import streams  # noqa: E402
from h264types import Sps, Pps, SliceHeader  # noqa: E402
from libh264_amd import gen  # noqa: E402


def mixed(name, cfg, seed0, n_slices, **kw):
    nals = streams.mixed_nals(cfg, n_slices=n_slices, seed0=seed0, **kw)
    rnd = random.Random(seed0)
    data = b"\x00\x00" + streams.annexb(
        nals, long_codes=[rnd.random() < 0.5 for _ in nals]) + b"\x00"
    rec = orc.parse_stream(data, want=orc.ORC_REC_WANT_T2)
    golden_io.save(os.path.join(HERE, name), data, rec)
    return len(data)


def damaged(name, cfg, seed0):
    nals = streams.damage(streams.mixed_nals(cfg, n_slices=16, seed0=seed0),
                          seed0 + 2)
    nals.insert(5, b"\x65")
    nals.insert(9, b"\x0c\xff\x80")
    nals.insert(12, b"\x18\x01\x02")
    rec = orc.parse_nalus(nals, want=orc.ORC_REC_WANT_T2)
    blob = b"".join(nals)
    sizes = np.array([len(n) for n in nals], dtype=np.uint32)
    golden_io.save(os.path.join(HERE, name), blob, rec, nal_sizes=sizes,
                   statuses=np.array(rec.statuses, dtype=np.int32))
    return len(blob)


def overflow(name, cfg, seed0, want_cases=8):
    """Damaged slices in which a coefficient lands OUTSIDE its block (index 15
    of an Intra16x16 AC / chroma AC block): the reference stores unchecked and
    goes on (src/h264_syntax_slice_data.h:222-226), so must oracle and kernel
    -- an earlier restatement stopped there with -EIO and delivered fewer
    macroblocks than the reference.  Found by a seeded search over bit flips of
    generated slices; kept: cases where the walk goes on for at least one more
    macroblock."""
    dropped = C.c_uint64.in_dll(orc.lib(), "orc_stat_dropped_coeffs")
    head = [gen.sps(cfg), gen.pps(cfg)]
    rnd = random.Random(seed0)
    keep = []
    tries = 0
    while len(keep) < want_cases and tries < 20000:
        tries += 1
        st = (2, 0, 1)[tries % 3]
        nal = bytearray(gen.slice_nal(cfg, streams.rich_slice(
            cfg, 1000 + tries % 97, st, pm_i16x16=600, density=9)))
        for _ in range(rnd.randrange(1, 4)):
            p = rnd.randrange(4, len(nal))
            nal[p] ^= 1 << rnd.randrange(8)
        before = dropped.value
        rec = orc.parse_nalus(head + [bytes(nal)], want=0)
        if dropped.value > before and rec.results[0].mb_count >= 4:
            keep.append(bytes(nal))
    assert len(keep) == want_cases, (len(keep), tries)
    nals = head + keep
    rec = orc.parse_nalus(nals, want=orc.ORC_REC_WANT_T2)
    blob = b"".join(nals)
    sizes = np.array([len(n) for n in nals], dtype=np.uint32)
    golden_io.save(os.path.join(HERE, name), blob, rec, nal_sizes=sizes,
                   statuses=np.array(rec.statuses, dtype=np.int32))
    return len(blob), tries


def ps_1080p(entropy):
    sps = Sps()
    sps.profile_idc = 100
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


SYNTH_GRID = [(8160, 0, 0, 0), (1, 0, 0, 0), (119, 1, -3, 1), (120, 0, 5, 2),
              (121, 7, 12, 0), (500, 4000, -20, 1), (3000, 120, 25, 2),
              (8160, 0, -26, 0), (64, 8096, 3, 3)]
SYNTH_KINDS = [(0, 7, 5, 3), (1, 5, 1, 2), (1, 1, 1, 0)]


def synth_cases(entropy):
    """(kind, slice_type, nal_type, ref_idc, header) in a fixed order."""
    for k, (mbc, first, qpd, idc) in enumerate(SYNTH_GRID):
        for kind, st, nal_type, ref in SYNTH_KINDS:
            sh = SliceHeader()
            sh.slice_type = st
            sh.first_mb_in_slice = first
            sh.frame_num = k % 16
            sh.slice_qp_delta = qpd
            sh.cabac_init_idc = idc
            sh.idr_pic_id = k
            if st % 5 == 1:
                sh.direct_spatial_mv_pred_flag = 1
            if st % 5 == 2 or entropy == 0:
                sh.cabac_init_idc = 0
            yield kind, mbc, nal_type, ref, sh


def synth(name):
    b = orc.Builder()
    blobs, sizes, digests = [], [], []
    for entropy in (0, 1):
        sps, pps = ps_1080p(entropy)
        for kind, mbc, nal_type, ref, sh in synth_cases(entropy):
            out = b.nalu(ref, nal_type, sps=sps, pps=pps, sh=sh,
                         kind=kind + 1, mb_count=mbc, cap=1 << 16)
            sizes.append(len(out))
            digests.append(hashlib.sha256(out).digest())
            if len(out) <= 600:
                blobs.append(out)
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"),
        sizes=np.array(sizes, dtype=np.uint32),
        sha256=np.frombuffer(b"".join(digests), dtype=np.uint8).reshape(-1, 32),
        small=np.frombuffer(b"".join(blobs), dtype=np.uint8))


def cabac(name):
    triples = [(st, idc, qp) for st in (0, 1, 2, 3, 4) for idc in (0, 1, 2, 3)
               for qp in range(-5, 60)]
    dig = [hashlib.sha256(orc.cabac_init_states(*t).tobytes()).digest()
           for t in triples]
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"),
        triples=np.array(triples, dtype=np.int32),
        sha256=np.frombuffer(b"".join(dig), dtype=np.uint8).reshape(-1, 32),
        qp26_I=orc.cabac_init_states(2, 0, 26))


def main():
    print("mixed_small", mixed("mixed_small", streams.small_cfg(), 0, 24))
    print("mixed_400", mixed("mixed_400", streams.small_cfg(cf=0), 50, 8))
    print("mixed_422_10", mixed("mixed_422_10", streams.small_cfg(cf=2, bd=2),
                                52, 8))
    print("mixed_444", mixed("mixed_444", streams.small_cfg(cf=3), 53, 8))
    # interlaced material (frame / top / bottom field pictures, no MBAFF),
    # pic_order_cnt_type 1, SP and SI slices beside I / P / B
    print("mixed_interlaced", mixed(
        "mixed_interlaced",
        streams.small_cfg(field_coding=1, pic_order_cnt_type=1,
                          bottom_field_pic_order_in_frame_present_flag=1),
        60, 10, types=(2, 0, 1, 3, 4)))
    print("damaged_small", damaged("damaged_small", streams.small_cfg(), 9))
    print("damaged_overflow", overflow("damaged_overflow", streams.small_cfg(),
                                       77))
    synth("synth_grid")
    cabac("cabac_init")
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
