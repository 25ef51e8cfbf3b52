"""ctypes binding of the synthetic stream generator (csrc/gen/h264gen.h)."""
import ctypes as C
import os

import numpy as np

from .ctypes_abi import GenCfg, GenSlice, t2_stride

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(HERE, "lib", "libh264gen.so")
        if not os.path.exists(path):
            from . import build
            build.build_gen()
        L = C.CDLL(path)
        L.h264gen_sps.argtypes = [C.POINTER(GenCfg), C.c_void_p, C.c_size_t]
        L.h264gen_pps.argtypes = [C.POINTER(GenCfg), C.c_void_p, C.c_size_t]
        L.h264gen_aud.argtypes = [C.c_uint32, C.c_void_p, C.c_size_t]
        L.h264gen_filler.argtypes = [C.c_uint32, C.c_void_p, C.c_size_t]
        L.h264gen_sei_user_data.argtypes = [C.c_uint32, C.c_uint64,
                                            C.c_void_p, C.c_size_t]
        L.h264gen_slice_nal.argtypes = [C.POINTER(GenCfg),
                                        C.POINTER(GenSlice), C.c_void_p,
                                        C.c_size_t, C.c_void_p, C.c_uint32]
        L.h264gen_batch.restype = C.c_long
        L.h264gen_batch.argtypes = [C.POINTER(GenCfg), C.c_void_p, C.c_uint32,
                                    C.c_int, C.c_void_p, C.c_size_t,
                                    C.c_void_p, C.c_void_p]
        L.h264gen_mb_to_slice_group_map.restype = C.c_long
        L.h264gen_mb_to_slice_group_map.argtypes = [
            C.POINTER(GenCfg), C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t]
        L.h264gen_stat_cabac_bins.restype = C.c_uint64
        L.h264gen_stat_cabac_bins.argtypes = [C.c_int]
        L.h264gen_slice_bound.restype = C.c_size_t
        L.h264gen_slice_bound.argtypes = [C.POINTER(GenCfg),
                                          C.POINTER(GenSlice)]
        _LIB = L
    return _LIB


def cabac_bins(reset=False):
    """bins coded by the generator's CABAC encoder since the last reset"""
    return int(lib().h264gen_stat_cabac_bins(1 if reset else 0))


def cfg_1080p(**kw):
    """Baseline-profile 1920x1088 CAVLC configuration of BASELINE C1/C2."""
    c = GenCfg(width_mbs=120, height_mbs=68, profile_idc=66, level_idc=40,
               chroma_format_idc=1, pic_order_cnt_type=2,
               max_num_ref_frames=1, direct_8x8_inference_flag=1,
               deblocking_filter_control_present_flag=1, crop_bottom=4)
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def cat_of(cfg):
    high = cfg.profile_idc in (100, 110, 122, 244, 44)
    return cfg.chroma_format_idc if high else 1


def _small(fn, *args, cap=1 << 16):
    buf = C.create_string_buffer(cap)
    r = fn(*args, buf, cap)
    if r < 0:
        raise OSError(-r, os.strerror(-r))
    return buf.raw[:r]


def sps(cfg):
    return _small(lib().h264gen_sps, C.byref(cfg))


def pps(cfg):
    return _small(lib().h264gen_pps, C.byref(cfg))


def aud(primary_pic_type=0):
    return _small(lib().h264gen_aud, primary_pic_type)


def filler(n):
    return _small(lib().h264gen_filler, n, cap=n + 64)


def sei_user_data(n, seed=1):
    return _small(lib().h264gen_sei_user_data, n, seed, cap=2 * n + 256)


def slice_nal(cfg, sl, want_t2=False):
    cap = lib().h264gen_slice_bound(C.byref(cfg), C.byref(sl))
    out = C.create_string_buffer(cap)
    t2 = None
    stride = 0
    if want_t2:
        stride = t2_stride(cat_of(cfg))
        t2 = (C.c_uint8 * (sl.num_mbs * stride))()
    r = lib().h264gen_slice_nal(C.byref(cfg), C.byref(sl), out, cap, t2,
                                stride)
    if r < 0:
        raise OSError(-r, os.strerror(-r))
    if want_t2:
        exp = np.frombuffer(bytes(t2), dtype=np.uint8).reshape(sl.num_mbs,
                                                               stride)
        return out.raw[:r], exp
    return out.raw[:r]


def slice_group_map(cfg, change_cycle=0, field_pic=0):
    """macroblock-to-slice-group map of a picture as the generator derives it
    from the standard (one uint8 per macroblock address)"""
    cap = 2 * cfg.width_mbs * cfg.height_mbs + 16
    buf = np.zeros(cap, dtype=np.uint8)
    n = lib().h264gen_mb_to_slice_group_map(
        C.byref(cfg), change_cycle, field_pic,
        buf.ctypes.data_as(C.c_void_p), cap)
    if n < 0:
        raise OSError(-n, os.strerror(-n))
    return buf[:n].copy()


def group_slices(cfg, change_cycle=0, field_pic=0, split=1):
    """(first_mb, num_mbs, addresses) for slices that together code every
    macroblock of the picture: each slice group in `split` pieces."""
    m = slice_group_map(cfg, change_cycle, field_pic)
    out = []
    for g in sorted(set(m.tolist())):
        addrs = np.nonzero(m == g)[0]
        for part in np.array_split(addrs, min(split, len(addrs))):
            if len(part):
                out.append((int(part[0]), len(part), part.astype(np.int64)))
    return out


def batch(cfg, slices, threads=None):
    """Generate len(slices) NAL units in parallel.  Returns (buf, offs, sizes)
    with buf a numpy uint8 array holding the NAL units at 16-byte aligned
    offsets."""
    n = len(slices)
    arr = (GenSlice * n)(*slices)
    cap = sum(lib().h264gen_slice_bound(C.byref(cfg), C.byref(s)) + 16
              for s in slices)
    buf = np.empty(cap, dtype=np.uint8)
    offs = np.zeros(n, dtype=np.uint64)
    sizes = np.zeros(n, dtype=np.uint32)
    if threads is None:
        threads = min(os.cpu_count() or 1, 32)
    r = lib().h264gen_batch(C.byref(cfg), arr, n, threads,
                            buf.ctypes.data_as(C.c_void_p), cap,
                            offs.ctypes.data_as(C.c_void_p),
                            sizes.ctypes.data_as(C.c_void_p))
    if r < 0:
        raise OSError(-r, os.strerror(-r))
    return buf[:r], offs, sizes
