"""Shared stream builders for the tests: seeded synthetic slices from the
generator (libh264_amd.gen) in a handful of configurations."""
import ctypes as C

import numpy as np

from libh264_amd import gen
from libh264_amd.ctypes_abi import GenCfg, GenSlice, t2_stride


def small_cfg(cf=1, bd=0, **kw):
    """A small picture with every coding tool the engine parses."""
    high = cf != 1 or bd != 0 or kw.get("transform_8x8_mode_flag", 1)
    c = GenCfg(width_mbs=8, height_mbs=6,
               profile_idc=(244 if cf == 3 else 122 if (cf == 2 or bd) else
                            100) if high else 66,
               level_idc=40, chroma_format_idc=cf, bit_depth_luma_minus8=bd,
               bit_depth_chroma_minus8=bd, pic_order_cnt_type=0,
               log2_max_poc_lsb_minus4=2, max_num_ref_frames=2,
               direct_8x8_inference_flag=1, transform_8x8_mode_flag=1,
               deblocking_filter_control_present_flag=1,
               num_ref_idx_l0_default_minus1=1,
               num_ref_idx_l1_default_minus1=0)
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def rich_slice(cfg, seed, st, first_mb=0, num_mbs=None, **kw):
    total = cfg.width_mbs * cfg.height_mbs
    if num_mbs is None:
        num_mbs = total - first_mb
    s = GenSlice(slice_type=st, idr=int(st % 5 == 2), nal_ref_idc=1,
                 first_mb=first_mb, num_mbs=num_mbs, frame_num=seed % 16,
                 poc_lsb=seed % 64, pm_i16x16=300, pm_pcm=30, pm_skip=200,
                 pm_intra_in_inter=150, pm_t8x8=400, pm_escape=20,
                 pm_zero_pcm=500, density=1 + seed % 12, pm_cbp_zero=100,
                 end_with_skip=seed % 2,
                 hdr_extras=(seed % 8) if st % 5 != 2 else 0,
                 num_ref_idx_l0_minus1=seed % 3,
                 num_ref_idx_l1_minus1=(seed // 3) % 3, seed=seed * 7 + st)
    for k, v in kw.items():
        setattr(s, k, v)
    return s


def pack_nals(nals, align=16, lead=0):
    """Concatenate NAL units into one buffer; returns (uint8 array, offsets,
    sizes).  `lead` shifts every NAL by that many bytes to exercise unaligned
    starts."""
    offs, sizes = [], []
    pos = 0
    chunks = []
    for n in nals:
        pad = (-pos) % align + lead
        chunks.append(b"\xff" * pad)
        pos += pad
        offs.append(pos)
        sizes.append(len(n))
        chunks.append(bytes(n))
        pos += len(n)
    chunks.append(b"\xff" * 64)
    return (np.frombuffer(b"".join(chunks), dtype=np.uint8).copy(), offs,
            sizes)


def annexb(nals, long_codes=None):
    """Annex-B byte stream; long_codes[i] selects a 4-byte start code."""
    out = bytearray()
    for i, n in enumerate(nals):
        four = True if long_codes is None else long_codes[i]
        out += b"\x00\x00\x00\x01" if four else b"\x00\x00\x01"
        out += bytes(n)
    return bytes(out)


def mixed_nals(cfg, n_slices=24, seed0=0, split=True, types=(2, 0, 1, 0),
               **slice_kw):
    """SPS, PPS, then pictures made of AUD + SEI + one or two slices, with a
    filler now and then and a repeated parameter-set pair in the middle.
    With cfg.field_coding pictures cycle frame / top field / bottom field."""
    nals = [gen.sps(cfg), gen.pps(cfg)]
    total = cfg.width_mbs * cfg.height_mbs
    for k in range(n_slices):
        seed = seed0 + k
        st = types[k % len(types)]
        kw = dict(slice_kw)
        if cfg.field_coding:
            kw["field_pic"] = (0, 1, 2, 1, 2)[k % 5]
            kw["frame_num"] = (k // 2) % 16
        if k % 3 == 0:
            nals.append(gen.aud(k % 3))
        if k % 5 == 0:
            nals.append(gen.sei_user_data(40 + k, seed + 1))
        if split and k % 2:
            half = total // 2
            nals.append(gen.slice_nal(
                cfg, rich_slice(cfg, seed, st, 0, half, **kw)))
            nals.append(gen.slice_nal(
                cfg, rich_slice(cfg, seed, st, half, total - half, **kw)))
        else:
            nals.append(gen.slice_nal(cfg, rich_slice(cfg, seed, st, **kw)))
        if k % 7 == 3:
            nals.append(gen.filler(20 + k))
        if k == n_slices // 2:
            nals += [gen.sps(cfg), gen.pps(cfg)]
    return nals


def damage(nals, seed, first=2):
    """Truncate / bit-flip / forbid some NAL units, deterministically."""
    import random
    rnd = random.Random(seed)
    out = [bytearray(n) for n in nals]
    for i in range(first, len(out)):
        r = rnd.random()
        if r < 0.15 and len(out[i]) > 8:
            del out[i][rnd.randrange(4, len(out[i])):]
        elif r < 0.30 and len(out[i]) > 8:
            p = rnd.randrange(1, len(out[i]))
            out[i][p] ^= 1 << rnd.randrange(8)
        elif r < 0.34:
            out[i][0] |= 0x80
    return [bytes(n) for n in out]
