"""The PRODUCT seen through its drop-in C API: ctypes bindings of
libh264_amd/lib/libh264.so and of tests/helpers/recorder.c (which drives that
library through <h264/h264.h> only and logs what the callbacks deliver)."""
import ctypes as C
import os
import subprocess

from h264types import (Sps, Pps, SliceHeader, NaluHeader, Ev, Rewrite,
                       Bitstream)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "libh264_amd", "lib")
_REC = None
_H264 = None

WANT_REWRITE = 2
WANT_DUMP = 4
READER_FLAGS_SLICE_DATA = 1


def h264():
    """libh264.so, built on demand (python -m libh264_amd.build)."""
    global _H264
    if _H264 is None:
        path = os.path.join(LIBDIR, "libh264.so")
        if not os.path.exists(path):
            from libh264_amd import build
            build.build_all()
        # the engine library first, globally, so that the rpath-less loaders
        # of the test helpers resolve it too
        C.CDLL(os.path.join(LIBDIR, "libh264_amd_engine.so"),
               mode=C.RTLD_GLOBAL)
        L = C.CDLL(path, mode=C.RTLD_GLOBAL)
        L.h264_ctx_new.argtypes = [C.POINTER(C.c_void_p)]
        L.h264_ctx_destroy.argtypes = [C.c_void_p]
        L.h264_ctx_clear.argtypes = [C.c_void_p]
        L.h264_ctx_clear_nalu.argtypes = [C.c_void_p]
        L.h264_ctx_set_nalu_header.argtypes = [C.c_void_p,
                                               C.POINTER(NaluHeader)]
        L.h264_ctx_set_sps.argtypes = [C.c_void_p, C.POINTER(Sps)]
        L.h264_ctx_set_pps.argtypes = [C.c_void_p, C.POINTER(Pps)]
        L.h264_ctx_set_slice_header.argtypes = [C.c_void_p,
                                                C.POINTER(SliceHeader)]
        L.h264_write_nalu.argtypes = [C.POINTER(Bitstream), C.c_void_p]
        L.h264_write_grey_i_slice.argtypes = [C.POINTER(Bitstream),
                                              C.c_void_p, C.c_uint32]
        L.h264_write_skipped_p_slice.argtypes = [C.POINTER(Bitstream),
                                                 C.c_void_p, C.c_uint32]
        L.h264_rewrite_slice_header.argtypes = [C.POINTER(Bitstream),
                                                C.c_void_p,
                                                C.POINTER(SliceHeader)]
        L.h264_parse_sps.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(Sps)]
        L.h264_parse_pps.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(Sps),
                                     C.POINTER(Pps)]
        L.h264_find_nalu.argtypes = [C.c_char_p, C.c_size_t,
                                     C.POINTER(C.c_size_t),
                                     C.POINTER(C.c_size_t)]
        _H264 = L
    return _H264


def recorder():
    global _REC
    if _REC is None:
        h264()
        src = os.path.join(ROOT, "tests", "helpers", "recorder.c")
        out_dir = os.path.join(ROOT, "tests", "helpers", "build")
        out = os.path.join(out_dir, "librecorder.so")
        deps = [src, os.path.join(ROOT, "oracle", "evlog.h")]
        if (not os.path.exists(out) or
                any(os.path.getmtime(d) > os.path.getmtime(out)
                    for d in deps)):
            os.makedirs(out_dir, exist_ok=True)
            subprocess.check_call([
                "gcc", "-std=gnu99", "-O1", "-g", "-fPIC", "-shared", "-Wall",
                "-I" + os.path.join(ROOT, "include"), src, "-o", out,
                "-L" + LIBDIR, "-lh264", "-Wl,-rpath," + LIBDIR])
        L = C.CDLL(out)
        L.rec_new.restype = C.c_void_p
        L.rec_new.argtypes = [C.c_uint32, C.c_uint32]
        L.rec_free.argtypes = [C.c_void_p]
        L.rec_parse_stream.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t,
                                       C.POINTER(C.c_size_t)]
        L.rec_parse_nalu.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_size_t]
        L.rec_ctx.restype = C.c_void_p
        L.rec_ctx.argtypes = [C.c_void_p]
        for name in ("events", "sps", "pps", "headers", "rewrites",
                     "rewrite_idx", "dumps"):
            f = getattr(L, "rec_" + name)
            f.restype = C.c_void_p
            f.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
        _REC = L
    return _REC


def _array(ptr, count, ctype):
    if not ptr or count == 0:
        return (ctype * 0)()
    return C.cast(ptr, C.POINTER(ctype * count)).contents


class Recording:
    def __init__(self, rec, keep):
        L = recorder()
        n = C.c_size_t()

        def grab(name, ctype):
            p = getattr(L, "rec_" + name)(rec, C.byref(n))
            arr = _array(p, n.value, ctype)
            return type(arr).from_buffer_copy(arr) if n.value else arr

        self.events = grab("events", Ev)
        self.sps = grab("sps", Sps)
        self.pps = grab("pps", Pps)
        self.headers = grab("headers", SliceHeader)
        self.rewrites = bytes(grab("rewrites", C.c_uint8))
        self.rewrite_idx = grab("rewrite_idx", Rewrite)
        self.dumps = bytes(grab("dumps", C.c_uint8)).decode()
        self.consumed = None
        self.result = None
        self._keep = keep

    def event_tuples(self):
        return [(e.kind, e.a, e.b, e.c) for e in self.events]


def parse_stream(data, flags=READER_FLAGS_SLICE_DATA, want=0, stop_after=0):
    """stop_after: as orc.parse_stream"""
    L = recorder()
    rec = L.rec_new(flags, want)
    off = C.c_size_t()
    buf = C.create_string_buffer(bytes(data), len(data))
    L.rec_stop_after.argtypes = [C.c_void_p, C.c_uint32]
    L.rec_stop_after.restype = None
    L.rec_stop_after(rec, stop_after)
    res = L.rec_parse_stream(rec, buf, len(data), C.byref(off))
    first = off.value
    if stop_after and res == 0 and first < len(data):
        L.rec_stop_after(rec, 0)
        res = L.rec_parse_stream(rec, C.c_char_p(C.addressof(buf) + first), len(data) - first,
                                 C.byref(off))
        off.value += first
    out = Recording(rec, buf)
    out.consumed = off.value
    out.consumed_first = first
    out.result = res
    L.rec_free(rec)
    return out


def parse_nalus(nalus, flags=READER_FLAGS_SLICE_DATA, want=0):
    L = recorder()
    rec = L.rec_new(flags, want)
    keep, statuses = [], []
    for n in nalus:
        b = C.create_string_buffer(bytes(n), len(n))
        keep.append(b)
        statuses.append(L.rec_parse_nalu(rec, C.addressof(b), C.addressof(b),
                                         len(n)))
    out = Recording(rec, keep)
    out.statuses = statuses
    L.rec_free(rec)
    return out
